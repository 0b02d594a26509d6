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
#include <algorithm>
#include <atomic>

#include "pha_ntt_core.h"

namespace pha {

void launch_add(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, hipStream_t s);

// carry-free split-accumulator MAC in base conversion (experiments library: pha_set_tuning key 1 switches it off)
#if defined(PHA_EXPERIMENTS)
std::atomic<int> g_bconv_split{1};
static inline bool bconv_split_on() { return g_bconv_split.load(std::memory_order_relaxed) != 0; }
#else
static constexpr bool bconv_split_on() { return true; }
#endif

constexpr int kBcThreads = 256;
constexpr int kBcMaxOutPerBlock = 32;  // output primes per workgroup (upper bound; the launch balances the groups)

struct BConvLaunch {
    const BConvDev *convs;       // device array
    uint32_t conv_step;          // converter of polynomial z = convs[z * conv_step]
    u64 *dst;                    // polynomial z: dst + z * dst_stride
    const u64 *src;              // polynomial z: src + z * src_stride (+ src_limb * n)
    const u64 *own;              // copy_own source (the NTT-form input of mod-up), same for every z
    size_t dst_stride, src_stride;
    const DModulus *mod;
    uint32_t n;
    uint32_t out_per_block;      // output primes per workgroup (<= kBcMaxOutPerBlock)
    // batches of ciphertexts (mod-up): polynomial z = (ciphertext z / conv_count, digit z % conv_count);
    // conv_count = 0: every z has its own converter index z (one ciphertext) or shares converter 0 (conv_step 0)
    uint32_t conv_count;
    size_t src_group_stride, own_group_stride;  // per ciphertext
    // optional epilogue (BFV mod-down, moddown_kernel rns_bconv.cu:680-689 + add_to_ct_kernel :763-769 fused into the conversion):
    // instead of storing delta_j the kernel stores epi_dst_j (+)= (epi_cx_j - delta_j) * epi_cst_j; null epi_cx = off
    const u64 *epi_cx;
    u64 *epi_dst;
    const u64x2 *epi_cst;        // [limb] constant with its Shoup quotient (P^-1 mod q_j)
    size_t epi_cx_stride, epi_dst_stride;
    uint32_t epi_acc;
    // optional bConv_HPS correction (BConvEpilogue::hps_inv / hps_alpha)
    const double *hps_inv;
    const u64 *hps_alpha;
};
struct BConvWho {
    uint32_t ci, grp;
};
__device__ __forceinline__ BConvWho bconv_who(const BConvLaunch &L) {
    const uint32_t z = blockIdx.z;
    return L.conv_count ? BConvWho{z % L.conv_count, z / L.conv_count} : BConvWho{z, 0u};
}

// bconv_mult (+) bconv_matmul (src/rns_bconv.cu:22-60,109-170; padded variant :455-485).
// SPLIT: carry-free MAC -- inputs are cut at bit SY and matrix entries at bit SM, the four partial
// products accumulate in plain 64-bit registers with one v_mad_u64_u32 each and are recombined once per
// output.  (SY, SM) = (30, 30): primes <= 60 bits, isz <= 16 (products < 2^60, 16 of them).  (30, 31) / (31, 30): isz <= 32,
// residues up to 60 bits on one side and up to 62 on the other (products < 2^61, recombined every 8 terms) -- the Q -> Bsk,
// Q -> R and back conversions of the BFV multiply, whose auxiliary primes are 61 bits wide (BConv::split_kind).
template <int ISZ_PAD, bool SCALE_IN, bool SPLIT, int SY = 30, int SM = 30>
__global__ __launch_bounds__(kBcThreads) void bconv_kernel(const BConvLaunch L) {
    constexpr int ROWPAD = ISZ_PAD > kBcRowPad ? 32 : kBcRowPad;
    const BConvWho who = bconv_who(L);
    const BConvDev &d = L.convs[who.ci * L.conv_step];
    const uint32_t coeff = blockIdx.x * kBcThreads + threadIdx.x;
    const uint32_t n = L.n;
    const u64 *src = L.src + (size_t)blockIdx.z * L.src_stride + (size_t)who.grp * L.src_group_stride + (size_t)d.src_limb * n;
    const u64 *own = L.own + (size_t)who.grp * L.own_group_stride;
    u64 *dst = L.dst + (size_t)blockIdx.z * L.dst_stride;
    const uint32_t isz = d.isz, osz = d.osz;
    const uint32_t j0 = blockIdx.y * L.out_per_block;
    const uint32_t j1 = min(j0 + L.out_per_block, osz);
    const bool mont = SPLIT && d.oninv != nullptr;
    const bool r90 = mont && SY == 30 && SM == 30 && d.r90 != 0;   // (uniform) word-wise REDC from the split accumulators
    // Everything the output loop needs per output prime goes through LDS once per workgroup: the matrix rows (SPLIT:
    // out_per_block x 16 entries of two dwords, read back as broadcast ds_read_b64) and the prime's constants and
    // destination limb.  (r02: the loop used to fetch oprime[j] and then mod[oprime[j]] from global memory for every
    // output -- two dependent vector loads per iteration, which made the kernel latency-bound at 2.3x its VALU time.)
    __shared__ uint2 s_rows[kBcMaxOutPerBlock * ROWPAD];
    __shared__ u64 s_p[kBcMaxOutPerBlock], s_c0[kBcMaxOutPerBlock], s_c1[kBcMaxOutPerBlock];
    __shared__ uint32_t s_jo[kBcMaxOutPerBlock];
    __shared__ u64x2 s_e[kBcMaxOutPerBlock];
    if (SPLIT) {
        // the converter's own row pitch may be smaller than this instantiation's (a short last digit next to full ones)
        const uint32_t pitch = d.row_pad;
        for (uint32_t e = threadIdx.x; e < L.out_per_block * ROWPAD; e += kBcThreads) {
            const uint32_t j = j0 + e / ROWPAD, i = e % ROWPAD;
            s_rows[e] = (j < osz && i < pitch) ? reinterpret_cast<const uint2 *>(d.mat30)[j * pitch + i] : uint2{0u, 0u};
        }
    }
    if (threadIdx.x < L.out_per_block && j0 + threadIdx.x < osz) {
        const uint32_t j = j0 + threadIdx.x;
        const DModulus m = L.mod[d.oprime[j]];
        s_p[threadIdx.x] = m.value;
        s_c0[threadIdx.x] = mont ? d.oninv[j] : m.ratio0;   // Montgomery: -p^-1 mod 2^64; Barrett: floor(2^128 / p)
        // r06 (2^90 rows): the 30-bit halves of p ride in the Barrett word the Montgomery branch does not use
        s_c1[threadIdx.x] = r90 ? ((m.value >> 30) << 32) | (m.value & 0x3fffffffu) : m.ratio1;
        const uint32_t jo = j + (j >= d.pad_start ? d.pad_len : 0);
        s_jo[threadIdx.x] = jo;
        if (L.epi_cx) s_e[threadIdx.x] = L.epi_cst[jo];
    }
    __syncthreads();
    if (j0 >= osz) return;  // (after the barrier) nothing to produce for this group
    u64 y[ISZ_PAD];
    u32 ylo[ISZ_PAD], yhi[ISZ_PAD];  // SPLIT: 30-bit halves, cut once per input
    double hps_frac = 0.0;           // bConv_HPS: sum_i y_i / q_i in the reference's order (hps_fix_kernel)
#pragma unroll
    for (int i = 0; i < ISZ_PAD; i++) {
        y[i] = 0;
        if (i < (int)isz) {
            u64 x = src[(size_t)i * n + coeff];
            if (SCALE_IN) x = shoup(x, d.hat_inv[i], L.mod[d.iprime[i]].value);
            y[i] = x;
            if (L.hps_inv) hps_frac = __builtin_fma((double)x, L.hps_inv[i], hps_frac);
        }
        ylo[i] = (u32)y[i] & ((1u << SY) - 1);
        yhi[i] = (u32)(y[i] >> SY);
    }
    if (d.copy_own && L.own && blockIdx.y == 0) {  // modup_copy_partQl_kernel rns_bconv.cu:522-528 (null own: the caller reads c2 itself)
        for (uint32_t i = 0; i < isz; i++)
            dst[(size_t)(d.src_limb + i) * n + coeff] = own[(size_t)(d.src_limb + i) * n + coeff];
    }
    const uint32_t count = j1 - j0;
    const u64 *hps_row = L.hps_inv ? L.hps_alpha + (size_t)llround(hps_frac) * osz + j0 : nullptr;
    for (uint32_t e = 0; e < count; e++) {
        u64 lo = 0, hi = 0, r = 0;
        bool reduced = false;
        if (SPLIT) {
            // rows are zero-padded to ROWPAD entries, and y[i] = 0 beyond isz: no per-term branch
            const uint2 *row = s_rows + e * ROWPAD;
            if (SY == 30 && SM == 30) {
                u64 ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
                for (int i = 0; i < ISZ_PAD; i++) {
                    const u32 y0 = ylo[i], y1 = yhi[i];
                    const uint2 mm = row[i];
                    const u32 m0 = mm.x, m1 = mm.y;
                    ll = (u64)y0 * m0 + ll;
                    lh = (u64)y0 * m1 + lh;
                    hl = (u64)y1 * m0 + hl;
                    hh = (u64)y1 * m1 + hh;
                }
                if (r90) {   // rows carry 2^90: three 30-bit REDC steps on the accumulators as they stand
                    const u64 pp = s_c1[e];
                    r = mont_redc90_split(ll, lh, hl, hh, s_p[e], (u32)pp, (u32)(pp >> 32), (u32)s_c0[e]);
                    reduced = true;
                } else {
                    // value = ll + (lh + hl) * 2^30 + hh * 2^60   (mid < 2^65: keep its carry)
                    const u64 mid = lh + hl;
                    const u64 mid_c = mid < lh ? 1 : 0;
                    lo = ll;
                    hi = 0;
                    const u64 t1 = mid << 30;
                    lo += t1;
                    hi += (lo < t1) + (mid >> 34) + (mid_c << 30);
                    const u64 t2 = hh << 60;
                    lo += t2;
                    hi += (lo < t2) + (hh >> 4);
                }
            } else {
                // halves of up to 30 / 31 bits: partial products below 2^61, so the four sums are folded into the 128-bit
                // total every 8 terms: value += ll + lh * 2^SM + hl * 2^SY + hh * 2^(SY + SM)
                lo = 0;
                hi = 0;
#pragma unroll
                for (int base = 0; base < ISZ_PAD; base += 8) {
                    u64 ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
                    for (int i = base; i < base + 8 && i < ISZ_PAD; i++) {
                        const u32 y0 = ylo[i], y1 = yhi[i];
                        const uint2 mm = row[i];
                        const u32 m0 = mm.x, m1 = mm.y;
                        ll = (u64)y0 * m0 + ll;
                        lh = (u64)y0 * m1 + lh;
                        hl = (u64)y1 * m0 + hl;
                        hh = (u64)y1 * m1 + hh;
                    }
                    lo += ll;
                    hi += lo < ll;
                    u64 t = lh << SM;
                    lo += t;
                    hi += (lo < t) + (lh >> (64 - SM));
                    t = hl << SY;
                    lo += t;
                    hi += (lo < t) + (hl >> (64 - SY));
                    t = hh << (SY + SM);
                    lo += t;
                    hi += (lo < t) + (hh >> (64 - SY - SM));
                }
            }
        } else {
            const u64 *row = d.mat + (size_t)(j0 + e) * isz;
            lo = 0;
            hi = 0;
#pragma unroll
            for (int i = 0; i < ISZ_PAD; i++)
                if (i < (int)isz) mac128(y[i], row[i], lo, hi);
        }
        // SPLIT rows are in Montgomery form when the converter has oninv (uniform): sum_i y_i * (qhat_i 2^64) < 16 * 2^60 * p
        // = 2^64 p, and REDC brings it to sum_i y_i * qhat_i mod p with a third of Barrett-128's multiplies
        const u64 p = s_p[e];
        if (reduced) {}
        else if (mont) r = mont_redc128(lo, hi, p, s_c0[e]);
        else r = barrett128(lo, hi, DModulus{p, s_c0[e], s_c1[e]});
        if (hps_row) r = sub_mod(r, hps_row[e], p);
        const size_t id = (size_t)s_jo[e] * n + coeff;
        if (L.epi_cx) {
            const u64 t = sub_mod(L.epi_cx[(size_t)blockIdx.z * L.epi_cx_stride + id], r, p);
            u64 o = shoup(t, s_e[e], p);
            u64 *out = L.epi_dst + (size_t)blockIdx.z * L.epi_dst_stride + id;
            if (L.epi_acc) o = add_mod(*out, o, p);
            *out = o;
        } else {
            dst[id] = r;
        }
    }
}

// generic fallback for very wide input bases (isz > 16): inputs are re-read per output prime
template <bool SCALE_IN>
__global__ __launch_bounds__(kBcThreads) void bconv_wide_kernel(const BConvLaunch L) {
    const BConvWho who = bconv_who(L);
    const BConvDev &d = L.convs[who.ci * L.conv_step];
    const uint32_t coeff = blockIdx.x * kBcThreads + threadIdx.x;
    const uint32_t n = L.n;
    const u64 *src = L.src + (size_t)blockIdx.z * L.src_stride + (size_t)who.grp * L.src_group_stride + (size_t)d.src_limb * n;
    const u64 *own = L.own + (size_t)who.grp * L.own_group_stride;
    u64 *dst = L.dst + (size_t)blockIdx.z * L.dst_stride;
    if (d.copy_own && L.own && blockIdx.y == 0) {
        for (uint32_t i = 0; i < d.isz; i++)
            dst[(size_t)(d.src_limb + i) * n + coeff] = own[(size_t)(d.src_limb + i) * n + coeff];
    }
    const uint32_t j0 = blockIdx.y * L.out_per_block;
    const uint32_t j1 = min(j0 + L.out_per_block, d.osz);
    for (uint32_t j = j0; j < j1; j++) {
        const DModulus m = L.mod[d.oprime[j]];
        const u64 *row = d.mat + (size_t)j * d.isz;
        u64 lo = 0, hi = 0;
        for (uint32_t i = 0; i < d.isz; i++) {
            u64 x = src[(size_t)i * n + coeff];
            if (SCALE_IN) x = shoup(x, d.hat_inv[i], L.mod[d.iprime[i]].value);
            mac128(x, row[i], lo, hi);
        }
        const uint32_t jo = j + (j >= d.pad_start ? d.pad_len : 0);
        dst[(size_t)jo * n + coeff] = barrett128(lo, hi, m);
    }
}

// Output groups (blockIdx.y) of a conversion launch: up to 32 outputs per workgroup (the inputs are loaded and scaled once per
// group: config 4's 30-output mod-down in one group, +4 % on that job) when that still leaves four workgroups per CU, 24 otherwise
// (a single polynomial at N = 2^15); balanced: 45 outputs -> 2 groups of 23
static uint32_t bconv_groups(const Context &c, uint32_t osz, uint32_t polys) {
    const uint32_t few = (osz + kBcMaxOutPerBlock - 1) / kBcMaxOutPerBlock;
    const uint32_t cap = (c.n / kBcThreads) * few * polys >= 1024 ? kBcMaxOutPerBlock : 24;
    return (osz + cap - 1) / cap;
}

// convs: device array; max_isz / max_osz over the converters used; split_kind: BConv::split_kind of the converters (0: none;
// a Tool passes its split_ok: 1 = every prime <= 60 bits)
void launch_bconv(Context &c, const BConvDev *convs, uint32_t conv_step, uint32_t batch, uint32_t max_isz,
                         uint32_t max_osz, int split_kind, u64 *dst, size_t dst_stride, const u64 *src,
                         size_t src_stride, const u64 *own, bool scale_in, hipStream_t s, uint32_t conv_count,
                         size_t group_stride, const BConvEpilogue *epi, size_t own_group_stride) {
    BConvLaunch L{};
    if (epi) {
        if (max_isz > 32) throw std::logic_error("the fused conversion epilogues need the register-resident converter");
        L.epi_cx = epi->cx; L.epi_dst = epi->dst; L.epi_cst = epi->cst;
        L.epi_cx_stride = epi->cx_stride; L.epi_dst_stride = epi->dst_stride; L.epi_acc = epi->accumulate ? 1 : 0;
        L.hps_inv = epi->hps_inv; L.hps_alpha = epi->hps_alpha;
    }
    L.conv_count = conv_count; L.src_group_stride = group_stride; L.own_group_stride = own_group_stride ? own_group_stride : group_stride;
    L.convs = convs; L.conv_step = conv_step; L.dst = dst; L.src = src; L.own = own;
    L.dst_stride = dst_stride; L.src_stride = src_stride; L.mod = c.d_mod.p; L.n = (uint32_t)c.n;
    const uint32_t groups = bconv_groups(c, max_osz, batch);
    L.out_per_block = (max_osz + groups - 1) / groups;
    dim3 grid((unsigned)(c.n / kBcThreads), groups, batch);
    dim3 block(kBcThreads);
    // the 30 / 30 split accumulators hold at most 16 terms; wider bases or wider primes take the 30 / 31 cuts or the 128-bit accumulate
    if (split_kind >= 2 && max_isz <= 32 && bconv_split_on()) {
#define PHA_BCW(P, SY, SM)                                                                                       \
    do {                                                                                                         \
        if (scale_in) hipLaunchKernelGGL((bconv_kernel<P, true, true, SY, SM>), grid, block, 0, s, L);            \
        else hipLaunchKernelGGL((bconv_kernel<P, false, true, SY, SM>), grid, block, 0, s, L);                    \
    } while (0)
        if (split_kind == 2) {
            if (max_isz <= 16) PHA_BCW(16, 30, 31);
            else PHA_BCW(32, 30, 31);
        } else {
            if (max_isz <= 16) PHA_BCW(16, 31, 30);
            else PHA_BCW(32, 31, 30);
        }
#undef PHA_BCW
        check_launch();
        return;
    }
    const bool split = split_kind == 1 && max_isz <= 16 && bconv_split_on();
#define PHA_BC(P)                                                                                        \
    do {                                                                                                 \
        if (scale_in && split) hipLaunchKernelGGL((bconv_kernel<P, true, true>), grid, block, 0, s, L);   \
        else if (scale_in) hipLaunchKernelGGL((bconv_kernel<P, true, false>), grid, block, 0, s, L);      \
        else if (split) hipLaunchKernelGGL((bconv_kernel<P, false, true>), grid, block, 0, s, L);         \
        else hipLaunchKernelGGL((bconv_kernel<P, false, false>), grid, block, 0, s, L);                   \
    } while (0)
    if (max_isz <= 2) PHA_BC(2);
    else if (max_isz <= 4) PHA_BC(4);
    else if (max_isz <= 8) PHA_BC(8);
    else if (max_isz == 15) PHA_BC(15);  // alpha = 15 (C3 / C4): no padded sixteenth term
    else if (max_isz <= 16) PHA_BC(16);
    else if (max_isz <= 32) PHA_BC(32);  // BFV multiply at 30-limb chains: inputs still live in registers
    else if (scale_in) hipLaunchKernelGGL((bconv_wide_kernel<true>), grid, block, 0, s, L);
    else hipLaunchKernelGGL((bconv_wide_kernel<false>), grid, block, 0, s, L);
#undef PHA_BC
    check_launch();
}

// ---- pha_keyswitch_rescale: the P -> Ql conversion of mod-down with the rescale's last-limb work folded in ----------------
// Per coefficient (one thread), with y_i the P residues of cx in coefficient form, already scaled by phat_i^-1 (the inverse
// transform's epilogue did that) and M' the converter whose rows carry P^-1 (Tool::p_to_ql_pinv):
//   dP_j   = sum_i y_i M'_ij mod q_j            = delta_j * P^-1        (moddown_from_NTT rns_bconv.cu:776-828 up to its NTT)
//   c_last = t_last - dP_last mod q_last          t_last = iNTT(ct_last + cx_last P^-1): coefficient form of the key-switched
//                                                 ciphertext's last limb (rns.cu:1171 after rns_bconv.cu:763-769)
//   v_j    = dP_j + (c_last mod q_j) mod q_j      j < Ql - 1
// One forward transform of v then gives out_j = (ct_j + cx_j P^-1 - NTT(v)_j) q_last^-1 (EPI_FWD_KSRESCALE), bit for bit what
// keyswitch_inplace followed by divide_and_round_q_last_ntt stores (NTT is linear; every stored value is canonical).
struct BConvRescaleLaunch {
    const BConvDev *conv;        // P -> Ql with P^-1 folded into the rows (Montgomery form)
    u64 *dst;                    // v: polynomial z at dst + z * dst_stride, [Ql - 1][N]
    const u64 *cx;               // polynomial z at cx + z * cx_stride: [QlP][N], limb Ql - 1 = t_last, limbs >= Ql = y
    size_t dst_stride, cx_stride;
    const DModulus *mod;
    uint32_t n, ql, out_per_block;
};
template <int ISZ_PAD>
__global__ __launch_bounds__(kBcThreads) void bconv_rescale_kernel(const BConvRescaleLaunch L) {
    const BConvDev &d = *L.conv;
    const uint32_t coeff = blockIdx.x * kBcThreads + threadIdx.x;
    const uint32_t n = L.n, nl = L.ql - 1;
    const u64 *cx = L.cx + (size_t)blockIdx.z * L.cx_stride;
    const u64 *src = cx + (size_t)L.ql * n;
    u64 *dst = L.dst + (size_t)blockIdx.z * L.dst_stride;
    const uint32_t j0 = blockIdx.y * L.out_per_block;
    const uint32_t j1 = min(j0 + L.out_per_block, nl);
    // rows / constants of this group's outputs, plus (slot out_per_block) those of the last data limb
    __shared__ uint2 s_rows[(kBcMaxOutPerBlock + 1) * kBcRowPad];
    __shared__ u64 s_p[kBcMaxOutPerBlock + 1], s_c0[kBcMaxOutPerBlock + 1], s_c1[kBcMaxOutPerBlock + 1];
    __shared__ uint2 s_pp[kBcMaxOutPerBlock + 1];   // r06: the 30-bit halves of p (mont_redc90_split)
    const bool r90 = d.r90 != 0;                    // (uniform) the rows carry 2^90
    for (uint32_t e = threadIdx.x; e < (L.out_per_block + 1) * kBcRowPad; e += kBcThreads) {
        const uint32_t slot = e / kBcRowPad, i = e % kBcRowPad;
        const uint32_t j = slot == L.out_per_block ? nl : j0 + slot;
        s_rows[e] = j <= nl ? reinterpret_cast<const uint2 *>(d.mat30)[j * kBcRowPad + i] : uint2{0u, 0u};
    }
    if (threadIdx.x <= L.out_per_block) {
        const uint32_t j = threadIdx.x == L.out_per_block ? nl : j0 + threadIdx.x;
        if (j <= nl) {
            const DModulus m = L.mod[d.oprime[j]];
            s_p[threadIdx.x] = m.value;
            s_c0[threadIdx.x] = d.oninv[j];
            s_c1[threadIdx.x] = m.ratio1;
            s_pp[threadIdx.x] = uint2{(uint32_t)(m.value & 0x3fffffffu), (uint32_t)(m.value >> 30)};
        }
    }
    __syncthreads();
    if (j0 >= nl) return;
    u32 ylo[ISZ_PAD], yhi[ISZ_PAD];
#pragma unroll
    for (int i = 0; i < ISZ_PAD; i++) {
        const u64 y = i < (int)d.isz ? src[(size_t)i * n + coeff] : 0;
        ylo[i] = (u32)y & 0x3fffffffu;
        yhi[i] = (u32)(y >> 30);
    }
    auto convert = [&](uint32_t slot) -> u64 {   // sum_i y_i * row_i, reduced (REDC) modulo the slot's prime
        const uint2 *row = s_rows + slot * kBcRowPad;
        u64 ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
        for (int i = 0; i < ISZ_PAD; i++) {
            const uint2 mm = row[i];
            ll = (u64)ylo[i] * mm.x + ll;
            lh = (u64)ylo[i] * mm.y + lh;
            hl = (u64)yhi[i] * mm.x + hl;
            hh = (u64)yhi[i] * mm.y + hh;
        }
        if (r90) {
            const uint2 pp = s_pp[slot];
            return mont_redc90_split(ll, lh, hl, hh, s_p[slot], pp.x, pp.y, (u32)s_c0[slot]);
        }
        const u64 mid = lh + hl, mid_c = mid < lh ? 1 : 0;
        u64 lo = ll, hi = 0;
        const u64 t1 = mid << 30;
        lo += t1;
        hi += (lo < t1) + (mid >> 34) + (mid_c << 30);
        const u64 t2 = hh << 60;
        lo += t2;
        hi += (lo < t2) + (hh >> 4);
        return mont_redc128(lo, hi, s_p[slot], s_c0[slot]);
    };
    const u64 q_last = s_p[L.out_per_block];
    const u64 c_last = sub_mod(cx[(size_t)nl * n + coeff], convert(L.out_per_block), q_last);
    const uint32_t count = j1 - j0;
    for (uint32_t e = 0; e < count; e++) {
        const u64 p = s_p[e];
        const u64 r = barrett64(c_last, p, s_c1[e]);   // divide_and_round_reduce_q_last_kernel rns.cu:1128-1139
        dst[(size_t)(j0 + e) * n + coeff] = add_mod(convert(e), r, p);
    }
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
    // pha_keyswitch_rescale: limb fix_limb of cx (the last data limb) receives ct_last + cx_last * P^-1 instead of cx_last
    // (its inverse transform is the first half of c_last; rns.cu:1171 after rns_bconv.cu:763-769); 0xffffffff = off
    uint32_t fix_limb;
    u64x2 fix_cst;           // P^-1 mod q_last
    const u64 *fix_ct;       // ct [2][Ql][N] of the (first) ciphertext
    size_t fix_ct_stride;    // Ql * N
    const FpInfo *fpinfo;    // [prime]: limbs below 2^50 form their sums in FP64 (batched kernel, r04)
    // r06 (batched kernel, NTT-form schemes): digit i's own limbs [i * alpha, min((i + 1) * alpha, ql)) are read from c2 where they lie
    // (own + b * own_stride) instead of from a verbatim copy inside t_mod_up, which the mod-up then does not make (modup's
    // own_in_place): 2 x 22.5 MiB per ciphertext at C3 that only moved data.  null = the copy is there (pha_modup's callers).
    const u64 *own;
    size_t own_stride;
    uint32_t alpha, ql;
};
__device__ __forceinline__ void inner_fix(const InnerArgs &k, uint32_t nid, const DModulus &m, size_t coeff, uint32_t b,
                                          u64x2 &r0, u64x2 &r1) {
    if (nid != k.fix_limb) return;   // uniform per workgroup
    const u64 *ct0 = k.fix_ct + (size_t)(2 * b) * k.fix_ct_stride + (size_t)nid * k.n + coeff;
    const u64x2 c0 = *reinterpret_cast<const u64x2 *>(ct0), c1 = *reinterpret_cast<const u64x2 *>(ct0 + k.fix_ct_stride);
    r0.x = add_mod(c0.x, shoup(r0.x, k.fix_cst, m.value), m.value);
    r0.y = add_mod(c0.y, shoup(r0.y, k.fix_cst, m.value), m.value);
    r1.x = add_mod(c1.x, shoup(r1.x, k.fix_cst, m.value), m.value);
    r1.y = add_mod(c1.y, shoup(r1.y, k.fix_cst, m.value), m.value);
}
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
    inner_fix(k, nid, m, coeff, 0, r0, r1);
    *reinterpret_cast<u64x2 *>(k.cx + c2_id) = r0;
    *reinterpret_cast<u64x2 *>(k.cx + c2_id + k.qlp_n) = r1;
}

// Batch of ciphertexts against ONE key: the key limbs are read once and stay in registers while the kernel
// walks the ciphertexts (the inner product is HBM-bound and 2/3 of its bytes are the key).
// r06: the digits are read once and the sums written once, in a launch that moves 157 MB per ciphertext: nontemporal loads and
// stores (33.6 -> 32.0 us per op at B = 32; see EW_TENSOR_NT in pha_poly.hip).
template <int BETA>
__global__ __launch_bounds__(256) void inner_prod_batched_kernel(const InnerArgs k, uint32_t batch) {
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t c2_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64x2 kb[BETA], ka[BETA];
#pragma unroll
    for (int i = 0; i < BETA; i++) {
        const u64 *key = k.evks[i];
        kb[i] = *reinterpret_cast<const u64x2 *>(key + evk_id);
        ka[i] = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
    }
    bool mine[BETA];   // (uniform) digit i holds this limb itself: its NTT form is c2's own
#pragma unroll
    for (int i = 0; i < BETA; i++) mine[i] = k.own != nullptr && nid < k.ql && nid >= (uint32_t)i * k.alpha && nid < ((uint32_t)i + 1) * k.alpha;
    if (k.fpinfo && k.fpinfo[twr].ok) {   // (uniform) r04: limbs below 2^50 -- every product an exact fp_mulmod_light (digits and key
        // words are canonical: a product is below 0.875 q, BETA <= 4 of them below 3.5 q), one fp_to_canon per sum: ~125 FP64
        // operations per ciphertext and thread where four 128-bit accumulators and four Barrett reductions take ~260
        const FpInfo fi = k.fpinfo[twr];
        const FpMod fm{fi.q, fi.qinv, false, false};
        double kbx[BETA], kby[BETA], kax[BETA], kay[BETA];
#pragma unroll
        for (int i = 0; i < BETA; i++) {
            kbx[i] = fp_from_canon(kb[i].x); kby[i] = fp_from_canon(kb[i].y);
            kax[i] = fp_from_canon(ka[i].x); kay[i] = fp_from_canon(ka[i].y);
        }
        for (uint32_t b = 0; b < batch; b++) {
            const u64 *mu = k.t_mod_up + (size_t)b * BETA * k.qlp_n + c2_id;
            const u64 *own = k.own + (size_t)b * k.own_stride + c2_id;
            u64 *cx = k.cx + (size_t)b * 2 * k.qlp_n + c2_id;
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                const u64x2 v = gload2<true>(reinterpret_cast<const u64x2 *>(mine[i] ? own : mu + (size_t)i * k.qlp_n));
                const double vx = fp_from_canon(v.x), vy = fp_from_canon(v.y);
                a0 += fp_mulmod_light(vx, kbx[i], fm);
                a1 += fp_mulmod_light(vy, kby[i], fm);
                b0 += fp_mulmod_light(vx, kax[i], fm);
                b1 += fp_mulmod_light(vy, kay[i], fm);
            }
            u64x2 r0{fp_to_canon(a0, fm), fp_to_canon(a1, fm)};
            u64x2 r1{fp_to_canon(b0, fm), fp_to_canon(b1, fm)};
            inner_fix(k, nid, m, coeff, b, r0, r1);
            gstore2<true>(reinterpret_cast<u64x2 *>(cx), r0);
            gstore2<true>(reinterpret_cast<u64x2 *>(cx + k.qlp_n), r1);
        }
        return;
    }
    for (uint32_t b = 0; b < batch; b++) {
        const u64 *mu = k.t_mod_up + (size_t)b * BETA * k.qlp_n + c2_id;
        const u64 *own = k.own + (size_t)b * k.own_stride + c2_id;
        u64 *cx = k.cx + (size_t)b * 2 * k.qlp_n + c2_id;
        u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
#pragma unroll
        for (int i = 0; i < BETA; i++) {
            const u64x2 v = gload2<true>(reinterpret_cast<const u64x2 *>(mine[i] ? own : mu + (size_t)i * k.qlp_n));
            mac128(v.x, kb[i].x, a0l, a0h);
            mac128(v.y, kb[i].y, a1l, a1h);
            mac128(v.x, ka[i].x, b0l, b0h);
            mac128(v.y, ka[i].y, b1l, b1h);
        }
        u64x2 r0{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
        u64x2 r1{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
        inner_fix(k, nid, m, coeff, b, r0, r1);
        gstore2<true>(reinterpret_cast<u64x2 *>(cx), r0);
        gstore2<true>(reinterpret_cast<u64x2 *>(cx + k.qlp_n), r1);
    }
}

// ---- hoisted rotations (src/evaluate.cu:1670-1866): for every output coefficient, sum over the Galois
//      elements e and digits b of  modup_b[perm_e[k]] * key_{e,b}[k].  The reference materialises the
//      permuted digits and adds per-element inner products; here the permutation is a gather inside ONE
//      inner-product kernel and the accumulation over elements stays in the 128-bit registers
//      (n_elts * beta * 2^120 < 2^128 needs n_elts * beta < 256; larger sets are split by the driver). ----
struct HoistArgs {
    u64 *cx;                          // [2][QlP][N]
    const u64 *t_mod_up;              // [beta][QlP][N]
    const u64 *const *const *keys;    // device array [n_elts] of device arrays [beta] of keys [2][QP][N]
    const uint32_t *const *tables;    // device array [n_elts] of NTT-domain permutation tables
    const DModulus *mod;
    const uint32_t *qlp_prime;
    uint32_t n, beta, n_elts, accumulate;  // accumulate: add to what cx already holds (split calls)
    size_t qlp_n, qp_n;
};
// BETA is a template parameter and every load of an element (2 x BETA gathered digit words, 4 x BETA key words, the next element's
// permutation entry) is issued before the first multiply (r03: the run-time digit loop serialised one memory round trip per digit)
template <int BETA>
__global__ __launch_bounds__(256) void hoist_inner_prod_kernel(const HoistArgs k) {
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t out_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
    if (k.accumulate) {
        const u64x2 p0 = *reinterpret_cast<const u64x2 *>(k.cx + out_id);
        const u64x2 p1 = *reinterpret_cast<const u64x2 *>(k.cx + out_id + k.qlp_n);
        a0l = p0.x; a1l = p0.y; b0l = p1.x; b1l = p1.y;
    }
    uint2 idx = *reinterpret_cast<const uint2 *>(k.tables[0] + coeff);
    for (uint32_t e = 0; e < k.n_elts; e++) {
        const uint2 idx_next = *reinterpret_cast<const uint2 *>(k.tables[e + 1 < k.n_elts ? e + 1 : e] + coeff);
        const u64 *const *keys = k.keys[e];
        u64 v0[BETA], v1[BETA];
        u64x2 kb[BETA], ka[BETA];
#pragma unroll
        for (int i = 0; i < BETA; i++) {
            const u64 *digit = k.t_mod_up + (size_t)i * k.qlp_n + (size_t)nid * k.n;
            const u64 *key = keys[i];
            v0[i] = digit[idx.x];
            v1[i] = digit[idx.y];
            kb[i] = *reinterpret_cast<const u64x2 *>(key + evk_id);
            ka[i] = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
        }
#pragma unroll
        for (int i = 0; i < BETA; i++) {
            mac128(v0[i], kb[i].x, a0l, a0h);
            mac128(v1[i], kb[i].y, a1l, a1h);
            mac128(v0[i], ka[i].x, b0l, b0h);
            mac128(v1[i], ka[i].y, b1l, b1h);
        }
        idx = idx_next;
    }
    *reinterpret_cast<u64x2 *>(k.cx + out_id) = u64x2{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
    *reinterpret_cast<u64x2 *>(k.cx + out_id + k.qlp_n) = u64x2{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
}
// any number of digits (run-time loop)
__global__ __launch_bounds__(256) void hoist_inner_prod_generic_kernel(const HoistArgs k) {
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t out_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
    if (k.accumulate) {
        const u64x2 p0 = *reinterpret_cast<const u64x2 *>(k.cx + out_id);
        const u64x2 p1 = *reinterpret_cast<const u64x2 *>(k.cx + out_id + k.qlp_n);
        a0l = p0.x; a1l = p0.y; b0l = p1.x; b1l = p1.y;
    }
    for (uint32_t e = 0; e < k.n_elts; e++) {
        const uint2 idx = *reinterpret_cast<const uint2 *>(k.tables[e] + coeff);
        const u64 *const *keys = k.keys[e];
        for (uint32_t i = 0; i < k.beta; i++) {
            const u64 *digit = k.t_mod_up + (size_t)i * k.qlp_n + (size_t)nid * k.n;
            const u64 v0 = digit[idx.x], v1 = digit[idx.y];
            const u64 *key = keys[i];
            const u64x2 kb = *reinterpret_cast<const u64x2 *>(key + evk_id);
            const u64x2 ka = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
            mac128(v0, kb.x, a0l, a0h);
            mac128(v1, kb.y, a1l, a1h);
            mac128(v0, ka.x, b0l, b0h);
            mac128(v1, ka.y, b1l, b1h);
        }
    }
    *reinterpret_cast<u64x2 *>(k.cx + out_id) = u64x2{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
    *reinterpret_cast<u64x2 *>(k.cx + out_id + k.qlp_n) = u64x2{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
}

// dst[limb][k] = sum_e src[limb][perm_e[k]] mod q  (c0 part of hoisting for ckks / bgv)
__global__ __launch_bounds__(256) void hoist_c0_kernel(u64 *dst, const u64 *src, const uint32_t *const *tables,
                                                       uint32_t n_elts, const DModulus *mod, uint32_t n) {
    const uint32_t limb = blockIdx.y;
    const u64 q = mod[limb].value;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    u64 acc = 0;
    for (uint32_t e = 0; e < n_elts; e++) acc = add_mod(acc, src[(size_t)limb * n + tables[e][coeff]], q);
    dst[(size_t)limb * n + coeff] = acc;
}

// ---- weighted hoisted rotations (BASELINE config 5, build-defined: no reference counterpart): as above with a
//      plaintext weight per Galois element, multiplied in before the shared mod-down:
//      cx[k] = sum_e w_e[k] * (sum_b modup_b[perm_e[k]] * key_{e,b}[k] mod q).  The per-element inner product
//      is reduced once (Barrett) so that the weighted sum fits the 128-bit accumulator again. ----
struct HoistWArgs {
    HoistArgs h;
    const u64 *const *weights;        // device array [n_elts] of weights [QlP][N] (NTT form)
};
template <int BETA>   // 0: run-time digit loop
__global__ __launch_bounds__(256) void hoist_weighted_inner_prod_kernel(const HoistWArgs kw) {
    const HoistArgs &k = kw.h;
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t out_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
    if (k.accumulate) {
        const u64x2 p0 = *reinterpret_cast<const u64x2 *>(k.cx + out_id);
        const u64x2 p1 = *reinterpret_cast<const u64x2 *>(k.cx + out_id + k.qlp_n);
        a0l = p0.x; a1l = p0.y; b0l = p1.x; b1l = p1.y;
    }
    uint2 idx = *reinterpret_cast<const uint2 *>(k.tables[0] + coeff);
    for (uint32_t e = 0; e < k.n_elts; e++) {
        const uint2 idx_next = *reinterpret_cast<const uint2 *>(k.tables[e + 1 < k.n_elts ? e + 1 : e] + coeff);
        const u64 *const *keys = k.keys[e];
        const u64x2 w = *reinterpret_cast<const u64x2 *>(kw.weights[e] + out_id);
        u64 s0l = 0, s0h = 0, s1l = 0, s1h = 0, t0l = 0, t0h = 0, t1l = 0, t1h = 0;
        if constexpr (BETA > 0) {   // every load of the element before the first multiply
            u64 v0[BETA], v1[BETA];
            u64x2 kb[BETA], ka[BETA];
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                const u64 *digit = k.t_mod_up + (size_t)i * k.qlp_n + (size_t)nid * k.n;
                const u64 *key = keys[i];
                v0[i] = digit[idx.x];
                v1[i] = digit[idx.y];
                kb[i] = *reinterpret_cast<const u64x2 *>(key + evk_id);
                ka[i] = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
            }
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                mac128(v0[i], kb[i].x, s0l, s0h);
                mac128(v1[i], kb[i].y, s1l, s1h);
                mac128(v0[i], ka[i].x, t0l, t0h);
                mac128(v1[i], ka[i].y, t1l, t1h);
            }
        } else {
            for (uint32_t i = 0; i < k.beta; i++) {
                const u64 *digit = k.t_mod_up + (size_t)i * k.qlp_n + (size_t)nid * k.n;
                const u64 v0 = digit[idx.x], v1 = digit[idx.y];
                const u64 *key = keys[i];
                const u64x2 kb = *reinterpret_cast<const u64x2 *>(key + evk_id);
                const u64x2 ka = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
                mac128(v0, kb.x, s0l, s0h);
                mac128(v1, kb.y, s1l, s1h);
                mac128(v0, ka.x, t0l, t0h);
                mac128(v1, ka.y, t1l, t1h);
            }
        }
        mac128(barrett128(s0l, s0h, m), w.x, a0l, a0h);
        mac128(barrett128(s1l, s1h, m), w.y, a1l, a1h);
        mac128(barrett128(t0l, t0h, m), w.x, b0l, b0h);
        mac128(barrett128(t1l, t1h, m), w.y, b1l, b1h);
        idx = idx_next;
    }
    *reinterpret_cast<u64x2 *>(k.cx + out_id) = u64x2{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
    *reinterpret_cast<u64x2 *>(k.cx + out_id + k.qlp_n) = u64x2{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
}

// dst[p][limb][k] = sum_e w_e[limb][k] * src[p][limb][perm_e[k]] mod q.  blockIdx.z = polynomial: c0 takes every
// element, c1 only the main-diagonal ones (Galois element 1, identity permutation), whose list starts at first_c1.
__global__ __launch_bounds__(256) void hoist_weighted_c_kernel(u64 *dst, const u64 *src, const uint32_t *const *tables,
                                                               const u64 *const *weights, uint32_t n_elts,
                                                               uint32_t first_c1, const DModulus *mod, uint32_t n,
                                                               size_t poly_stride) {
    const uint32_t limb = blockIdx.y, p = blockIdx.z;
    const DModulus m = mod[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)limb * n + coeff;
    u64 lo = 0, hi = 0;
    uint32_t terms = 0;
    for (uint32_t e = p ? first_c1 : 0; e < n_elts; e++) {
        mac128(src[p * poly_stride + (size_t)limb * n + tables[e][coeff]], weights[e][id], lo, hi);
        if (++terms == 48) {           // products of two 61-bit residues: 48 * 2^122 stays below 2^128
            lo = barrett128(lo, hi, m);
            hi = 0;
            terms = 1;
        }
    }
    dst[p * poly_stride + id] = barrett128(lo, hi, m);
}

// CPT consecutive words as one access
template <int CPT>
__device__ __forceinline__ void load_words(const u64 *p, u64 (&v)[CPT]) {
    if constexpr (CPT == 2) {
        const u64x2 t = *reinterpret_cast<const u64x2 *>(p);
        v[0] = t.x;
        v[1] = t.y;
    } else {
        v[0] = *p;
    }
}
template <int CPT>
__device__ __forceinline__ void store_words(u64 *p, const u64 (&v)[CPT]) {
    if constexpr (CPT == 2) *reinterpret_cast<u64x2 *>(p) = u64x2{v[0], v[1]};
    else *p = v[0];
}

// ---- baby-step / giant-step form of the weighted hoisted rotations (BASELINE config 5, build-defined) -----------------------
//      out = sum_i rot_{G_i}( sum_j w_ij (.) rot_{B_j}(ct) ):  d = ng * nb diagonals from nb - 1 baby keys and ng - 1 giant keys
//      instead of d - 1 keys (24 GB of Galois keys per 128-diagonal block at C3 -> 4 GB).  "Double hoisting": the baby rotations
//      share ONE mod-up of c1 and their inner products stay in the extended base [Q_l || P]; every giant step i weights them with
//      its own plaintexts w_ij (given over [Q_l || P], as in pha_hoisting_weighted) and pays one mod-down; the giant rotations
//      share ONE final mod-down of the sum of their inner products.
// One thread = one coefficient of one limb; the accumulators of NG giant steps live in registers, so the baby keys, the gathered
// digits and the per-baby Barrett reductions are paid once for all of them.
struct BsgsArgs {
    u64 *acc;                         // [ng][2][QlP][N]
    const u64 *t_mod_up;              // [beta][QlP][N]
    const u64 *const *const *keys;    // device [nb]: key table of baby j ([beta] keys), null for the identity
    const uint32_t *const *tables;    // device [nb]: NTT-domain permutation of baby j
    const u64 *const *weights;        // device [ng][nb]: w_ij over [QlP][N], null = no such term
    const DModulus *mod;
    const uint32_t *qlp_prime;
    uint32_t n, beta, nb, g0;         // g0: first giant step of this launch
    size_t qlp_n, qp_n;
    const u64 *cc;                    // the input ciphertext (c0, c1), [2][Ql][N]
    const u64x2 *p_mod_q;             // [Ql] P mod q_j with its Shoup quotient
    uint32_t ql;
    size_t ql_n;
    const FpInfo *fpinfo;             // [prime]: limbs below 2^50 accumulate in doubles (r04)
};
// The c0 / c1 terms ride in the same accumulators: on a data limb j the value added is P * x mod q_j, which the mod-down that
// follows (it divides by P exactly: (cx_j - conv(cx_P)_j) P^-1) turns back into x, and on the P limbs P * x = 0 -- so
// moddown(acc + P * y) = moddown(acc) + y, word for word what a separate weighted sum over Q_l would add afterwards.
// Every load of a baby step (3 gathered digits, the c0 word, 2 x BETA key words, NG weights, the next step's permutation
// entry) is issued before the first multiply: the first version branched per giant step on a null weight and looped over the
// digits at run time, which serialised ~12 memory round trips per baby step (2.9 TB/s); missing weights now point at a zero
// plane supplied by the driver.
// r04: limbs whose prime is below 2^50 run the same sums in FP64 (pha_arith.h: residues as doubles with integer values, every
// product an exact fp_mulmod_light).  A 128-bit multiply-accumulate costs ~12 vector instructions on 32-bit halves and the two
// Barrett reductions per baby step ~55; here a product-and-add is 7 FP64 operations and there is nothing to reduce at the end of a
// step but two re-centrings: ~210 instead of ~340 instructions per baby step at NG = 8, and 2 NG instead of 4 NG accumulator
// register pairs.  Magnitudes: gathered digits, key words and weights are canonical (< q), so a digit product is below 0.875 q and
// BETA + 1 of them below 4.4 q; s and t are re-centred (<= q/2), a weighted term is then below 0.69 q, and the accumulators are
// re-centred every 8 baby steps (0.5 + 8 x 0.69 = 6.0 q < 8 q = 2^53): every value is an exact integer, the stored residues are
// the ones the integer form stores.
template <int NG, int BETA>
__device__ __forceinline__ void hoist_bsgs_body_fp(const BsgsArgs &k, uint32_t nid, uint32_t twr, const FpInfo fi) {
    const FpMod fm{fi.q, fi.qinv, false, false};
    const size_t coeff = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t out_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    const bool data_limb = nid < k.ql;          // uniform
    const double pqd = data_limb ? fp_from_canon(k.p_mod_q[nid].x) : 0.0;
    const u64 *cc0 = k.cc + (size_t)(data_limb ? nid : 0) * k.n;
    double al[NG], bl[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) al[g] = bl[g] = 0.0;
    uint32_t idx = k.tables[0][coeff];
    for (uint32_t j = 0; j < k.nb; j++) {
        const u64 *const *keys = k.keys[j];
        const uint32_t idx_next = k.tables[j + 1 < k.nb ? j + 1 : j][coeff];
        u64 wv[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) wv[g] = k.weights[(size_t)(k.g0 + g) * k.nb + j][out_id];
        const u64 x0 = cc0[idx];
        double s, t;
        if (keys) {   // uniform
            u64 v[BETA], kb[BETA], ka[BETA];
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                const u64 *key = keys[i];
                v[i] = k.t_mod_up[(size_t)i * k.qlp_n + (size_t)nid * k.n + idx];
                kb[i] = key[evk_id];
                ka[i] = key[evk_id + k.qp_n];
            }
            s = data_limb ? fp_mulmod_light(fp_from_canon(x0), pqd, fm) : 0.0;   // + P * rot_j(c0)
            t = 0.0;
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                const double vd = fp_from_canon(v[i]);
                s += fp_mulmod_light(vd, fp_from_canon(kb[i]), fm);
                t += fp_mulmod_light(vd, fp_from_canon(ka[i]), fm);
            }
        } else {      // identity baby step: (P c0, P c1) on the data limbs, nothing on the P limbs
            s = data_limb ? fp_mulmod_light(fp_from_canon(x0), pqd, fm) : 0.0;
            t = data_limb ? fp_mulmod_light(fp_from_canon(k.cc[k.ql_n + (size_t)nid * k.n + idx]), pqd, fm) : 0.0;
        }
        s = fp_reduce(s, fm);
        t = fp_reduce(t, fm);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const double wd = fp_from_canon(wv[g]);
            al[g] += fp_mulmod_light(s, wd, fm);
            bl[g] += fp_mulmod_light(t, wd, fm);
        }
        if ((j & 7u) == 7u) {   // (uniform)
#pragma unroll
            for (int g = 0; g < NG; g++) {
                al[g] = fp_reduce(al[g], fm);
                bl[g] = fp_reduce(bl[g], fm);
            }
        }
        idx = idx_next;
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        u64 *acc = k.acc + (size_t)(k.g0 + g) * 2 * k.qlp_n + out_id;
        acc[0] = fp_to_canon(al[g], fm);
        acc[k.qlp_n] = fp_to_canon(bl[g], fm);
    }
}

// integer limbs: accumulators [goff, goff + NG) of the launch
template <int NG, int BETA>
__device__ __forceinline__ void hoist_bsgs_body_int(const BsgsArgs &k, uint32_t nid, uint32_t twr, uint32_t goff) {
    const DModulus m = k.mod[twr];
    const size_t coeff = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t out_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    const bool data_limb = nid < k.ql;          // uniform
    const u64x2 pq = data_limb ? k.p_mod_q[nid] : u64x2{0, 0};
    const u64 *cc0 = k.cc + (size_t)(data_limb ? nid : 0) * k.n;   // (P limbs: any valid row, the value is multiplied by 0)
    u64 al[NG], ah[NG], bl[NG], bh[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) al[g] = ah[g] = bl[g] = bh[g] = 0;
    uint32_t idx = k.tables[0][coeff];
    for (uint32_t j = 0; j < k.nb; j++) {
        const u64 *const *keys = k.keys[j];
        const uint32_t idx_next = k.tables[j + 1 < k.nb ? j + 1 : j][coeff];
        // loads first
        u64 wv[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) wv[g] = k.weights[(size_t)(k.g0 + goff + g) * k.nb + j][out_id];
        const u64 x0 = cc0[idx];
        u64 s, t;
        if (keys) {   // uniform
            u64 v[BETA], kb[BETA], ka[BETA];
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                const u64 *key = keys[i];
                v[i] = k.t_mod_up[(size_t)i * k.qlp_n + (size_t)nid * k.n + idx];
                kb[i] = key[evk_id];
                ka[i] = key[evk_id + k.qp_n];
            }
            u64 sl = 0, sh = 0, tl = 0, th = 0;
#pragma unroll
            for (int i = 0; i < BETA; i++) {
                mac128(v[i], kb[i], sl, sh);
                mac128(v[i], ka[i], tl, th);
            }
            s = barrett128(sl, sh, m);
            t = barrett128(tl, th, m);
            if (data_limb) s = add_mod(s, shoup(x0, pq, m.value), m.value);          // + P * rot_j(c0)
        } else {      // identity baby step: (P c0, P c1) on the data limbs, nothing on the P limbs
            s = data_limb ? shoup(x0, pq, m.value) : 0;
            t = data_limb ? shoup(k.cc[k.ql_n + (size_t)nid * k.n + idx], pq, m.value) : 0;
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            mac128(s, wv[g], al[g], ah[g]);
            mac128(t, wv[g], bl[g], bh[g]);
        }
        idx = idx_next;
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        u64 *acc = k.acc + (size_t)(k.g0 + goff + g) * 2 * k.qlp_n + out_id;
        acc[0] = barrett128(al[g], ah[g], m);
        acc[k.qlp_n] = barrett128(bl[g], bh[g], m);
    }
}

// NG accumulators per launch.  NG = 16 (r04): the FP64 limbs keep 16 (block, giant step) accumulators in 32 register pairs, so the
// baby keys are streamed once per 16 / ng row blocks; an integer limb would need 64 pairs for that and walks its baby steps twice
// with 8 accumulators each instead (16 of 60 limbs at the C3 set).
template <int NG, int BETA>
__global__ __launch_bounds__(256) void hoist_bsgs_inner_prod_kernel(const BsgsArgs k) {
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    if (k.fpinfo) {   // (uniform) FP64 limbs
        const FpInfo fi = k.fpinfo[twr];
        if (fi.ok) {
            hoist_bsgs_body_fp<NG, BETA>(k, nid, twr, fi);
            return;
        }
    }
    if constexpr (NG > 8) {
        hoist_bsgs_body_int<8, BETA>(k, nid, twr, 0);
        hoist_bsgs_body_int<NG - 8, BETA>(k, nid, twr, 8);
    } else {
        hoist_bsgs_body_int<NG, BETA>(k, nid, twr, 0);
    }
}

// giant steps, part 1: ct0 = sum_i B_i0[perm_Gi], ct1 = sum over the identity giant steps of B_i1, and the dense operands
// g1[z] = B_i1[perm_Gi] of the keyed giant steps (z = their rank among the keyed ones)
// blockIdx.z = row block: its ng giant steps start at B + z * ng * 2 polynomials, its nk operands at g1 + z * nk, its output at ct + 2 z
__global__ __launch_bounds__(256) void bsgs_combine_kernel(u64 *ct, u64 *g1, const u64 *B, const uint32_t *const *tables,
                                                           const uint32_t *keyed_rank, uint32_t ng, uint32_t nk, const DModulus *mod,
                                                           uint32_t n, size_t poly_stride) {
    const uint32_t limb = blockIdx.y;
    ct += (size_t)blockIdx.z * 2 * poly_stride;
    g1 += (size_t)blockIdx.z * nk * poly_stride;
    B += (size_t)blockIdx.z * ng * 2 * poly_stride;
    const u64 q = mod[limb].value;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)limb * n + coeff;
    u64 r0 = 0, r1 = 0;
    for (uint32_t i = 0; i < ng; i++) {
        const uint32_t from = tables[i][coeff];
        const u64 *b = B + (size_t)(2 * i) * poly_stride + (size_t)limb * n;
        r0 = add_mod(r0, b[from], q);
        const uint32_t z = keyed_rank[i];
        if (z == 0xffffffffu) r1 = add_mod(r1, b[poly_stride + coeff], q);     // identity giant step
        else g1[(size_t)z * poly_stride + id] = b[poly_stride + from];
    }
    ct[id] = r0;
    ct[poly_stride + id] = r1;
}

// giant steps, part 2: cx = sum_z <modup(g1[z]), key_z> in one pass (the inner products of nk key switches that share a mod-down)
struct MultiInnerArgs {
    u64 *cx;                        // [2][QlP][N]
    const u64 *t_mod_up;            // [nk][beta][QlP][N]
    const u64 *const *const *keys;  // device [nk] -> [beta]
    const DModulus *mod;
    const uint32_t *qlp_prime;
    uint32_t n, beta, nk;
    size_t qlp_n, qp_n;
};
__global__ __launch_bounds__(256) void inner_prod_multi_kernel(const MultiInnerArgs kk) {   // blockIdx.z = row block
    MultiInnerArgs k = kk;
    k.cx += (size_t)blockIdx.z * 2 * k.qlp_n;
    k.t_mod_up += (size_t)blockIdx.z * k.nk * k.beta * k.qlp_n;
    const uint32_t nid = blockIdx.y;
    const uint32_t twr = k.qlp_prime[nid];
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t c2_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
    for (uint32_t z = 0; z < k.nk; z++) {
        const u64 *const *keys = k.keys[z];
        for (uint32_t i = 0; i < k.beta; i++) {
            const u64 *key = keys[i];
            const u64x2 v = *reinterpret_cast<const u64x2 *>(k.t_mod_up + ((size_t)z * k.beta + i) * k.qlp_n + c2_id);
            const u64x2 kb = *reinterpret_cast<const u64x2 *>(key + evk_id);
            const u64x2 ka = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
            mac128(v.x, kb.x, a0l, a0h);
            mac128(v.y, kb.y, a1l, a1h);
            mac128(v.x, ka.x, b0l, b0h);
            mac128(v.y, ka.y, b1l, b1h);
        }
    }
    *reinterpret_cast<u64x2 *>(k.cx + c2_id) = u64x2{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
    *reinterpret_cast<u64x2 *>(k.cx + c2_id + k.qlp_n) = u64x2{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
}

// ---- (cx - delta) * c element-wise: moddown_kernel rns_bconv.cu:680-689 and
//      divide_and_round_q_last_kernel rns.cu:1082-1108 (REDUCE_LAST) ---------------------------------
struct SubMulArgs {
    u64 *dst;
    const u64 *cx, *delta;
    const u64x2 *cst;  // per-limb constant (Shoup pair)
    const DModulus *mod;
    uint32_t n;
    // polynomial blockIdx.z of a batch: element strides (0 for a single polynomial)
    size_t dst_stride = 0, cx_stride = 0, delta_stride = 0;
    uint32_t accumulate = 0;   // dst += result (the add_to_ct_kernel of keyswitch_inplace, rns_bconv.cu:763-769, fused)
};
template <bool REDUCE_LAST>
__global__ __launch_bounds__(256) void sub_mul_kernel(const SubMulArgs k) {
    const uint32_t limb = blockIdx.y;
    const DModulus m = k.mod[limb];
    const u64x2 cst = k.cst[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const u64 *delta = k.delta + (size_t)blockIdx.z * k.delta_stride;
    const u64 *cx = k.cx + (size_t)blockIdx.z * k.cx_stride;
    u64 *dst = k.dst + (size_t)blockIdx.z * k.dst_stride;
    u64 d = REDUCE_LAST ? barrett64(delta[coeff], m.value, m.ratio1) : delta[(size_t)limb * k.n + coeff];
    const u64 t = sub_mod(cx[(size_t)limb * k.n + coeff], d, m.value);
    u64 r = shoup(t, cst, m.value);
    if (k.accumulate) r = add_mod(dst[(size_t)limb * k.n + coeff], r, m.value);
    dst[(size_t)limb * k.n + coeff] = r;
}

// ---- BGV (plain modulus t): base_P_to_t_conv (rns.cu:283) as one thread per coefficient, writing over
//      the first P limb it has just read; bgv_moddown_kernel rns_bconv.cu:636-652;
//      bgv_mod_t_divide_q_kernel rns.cu:1186-1208 ---------------------------------------------------
struct PToTArgs {
    u64 *cx_p;                 // P block of polynomial 0: [alpha][N]; limb 0 receives [.]_t
    size_t stride;             // elements between polynomials (blockIdx.y)
    const u64x2 *hat_inv;      // [alpha] phat_i^-1 mod p_i
    const uint32_t *iprime;    // [alpha] rows of the QP table
    const u64 *hat_mod_t;      // [alpha] phat_i mod t
    const DModulus *mod;
    DModulus t;
    uint32_t isz, n;
};
__global__ __launch_bounds__(256) void p_to_t_kernel(const PToTArgs k) {
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    u64 *p = k.cx_p + (size_t)blockIdx.y * k.stride + coeff;
    u64 lo = 0, hi = 0;
    for (uint32_t i = 0; i < k.isz; i++) {
        const u64 x = shoup(p[(size_t)i * k.n], k.hat_inv[i], k.mod[k.iprime[i]].value);
        mac128(x, k.hat_mod_t[i], lo, hi);
    }
    p[0] = barrett128(lo, hi, k.t);
}

struct BgvDownArgs {
    u64 *dst;
    const u64 *cx, *delta, *cp_t;
    const u64x2 *p_mod_q, *pinv;
    const DModulus *mod;
    u64x2 pinv_t;
    u64 t;
    uint32_t n;
    size_t dst_stride = 0, cx_stride = 0, delta_stride = 0;   // polynomial blockIdx.z of a batch (cp_t moves with cx)
};
__global__ __launch_bounds__(256) void bgv_moddown_kernel(const BgvDownArgs k) {
    const uint32_t limb = blockIdx.y;
    const u64 q = k.mod[limb].value;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)limb * k.n + coeff;
    const size_t z = blockIdx.z;
    const u64 u = shoup(k.cp_t[z * k.cx_stride + coeff], k.pinv_t, k.t);
    const u64 corr = shoup(u, k.p_mod_q[limb], q);
    const u64 d = add_mod(sub_mod(k.cx[z * k.cx_stride + id], k.delta[z * k.delta_stride + id], q), corr, q);
    k.dst[z * k.dst_stride + id] = shoup(d, k.pinv[limb], q);
}

struct BgvSwitchArgs {
    u64 *dst;
    const u64 *src;            // polynomial 0: [ql][N] in coefficient form
    const u64x2 *q_last_mod_q, *inv_q_last;
    const DModulus *mod;
    u64x2 inv_q_last_t;
    DModulus t;
    uint32_t n, nl;
    size_t src_stride, dst_stride;
};
__global__ __launch_bounds__(256) void bgv_switch_kernel(const BgvSwitchArgs k) {
    const uint32_t limb = blockIdx.y;
    const DModulus m = k.mod[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const u64 *in = k.src + (size_t)blockIdx.z * k.src_stride;
    const u64 last = in[(size_t)k.nl * k.n + coeff];
    const u64 delta = barrett64(last, m.value, m.ratio1);
    const u64 last_t = barrett64(last, k.t.value, k.t.ratio1);
    const u64 u = shoup(last_t, k.inv_q_last_t, k.t.value);
    const u64 corr = shoup(u, k.q_last_mod_q[limb], m.value);
    const u64 d = add_mod(sub_mod(in[(size_t)limb * k.n + coeff], delta, m.value), corr, m.value);
    k.dst[(size_t)blockIdx.z * k.dst_stride + (size_t)limb * k.n + coeff] = shoup(d, k.inv_q_last[limb], m.value);
}

// ---- key-switching key generation, arithmetic part (src/secretkey.cu:232-341): per digit d and limb j of QP
//      b = -(a*s + u) [multiply_and_add_negate_rns_poly polymath.cu:234-251], and on the digit's own limbs
//      b += P * new_key [multiply_temp_mod_and_add_rns_poly polymath.cu:318-338]; the key is (b, a). ----
struct KeyGenArgs {
    u64 *const *evk;        // device array [dnum] of keys [2][QP][N]
    const u64 *sk;          // [QP][N] NTT form
    const u64 *new_key;     // [Q][N] NTT form
    const u64 *a, *u;       // [dnum][QP][N]: uniform part, noise in NTT form
    const u64x2 *p_mod_q;   // [Q]
    const DModulus *mod;
    uint32_t n, size_qp, alpha;
};
__global__ __launch_bounds__(256) void kswitch_key_kernel(const KeyGenArgs k) {
    const uint32_t limb = blockIdx.y, d = blockIdx.z;
    const DModulus m = k.mod[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)limb * k.n + coeff, did = (size_t)d * k.size_qp * k.n + id;
    const u64 a = k.a[did];
    u64 b = neg_mod(add_mod(mul_mod(a, k.sk[id], m), k.u[did], m.value), m.value);
    if (limb >= d * k.alpha && limb < (d + 1) * k.alpha)
        b = add_mod(b, shoup(k.new_key[id], k.p_mod_q[limb], m.value), m.value);
    u64 *key = k.evk[d];
    key[id] = b;
    key[(size_t)k.size_qp * k.n + id] = a;
}
__global__ __launch_bounds__(256) void scale_by_t_kernel(u64 *e, const DModulus *mod, u64 t, uint32_t n, size_t stride) {
    const DModulus m = mod[blockIdx.y];
    const size_t id = (size_t)blockIdx.z * stride + (size_t)blockIdx.y * n + blockIdx.x * 256 + threadIdx.x;
    e[id] = mul_mod(e[id], barrett64(t, m.value, m.ratio1), m);
}

// ---- Galois (src/galois.cu:11-39) ----------------------------------------------------------------
// elt^-1 mod 2n for an odd elt (Newton: x <- x (2 - elt x) doubles the correct low bits)
static uint32_t inv_mod_2n(uint32_t elt, size_t n) {
    uint32_t x = elt;                      // correct to 3 bits: elt^2 = 1 mod 8
    for (int i = 0; i < 5; i++) x *= 2 - elt * x;
    return x & (uint32_t)(2 * n - 1);
}
__global__ __launch_bounds__(256) void galois_ntt_kernel(u64 *dst, const u64 *src, const uint32_t *table, uint32_t n) {
    const uint32_t limb = blockIdx.y;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    dst[(size_t)limb * n + coeff] = src[(size_t)limb * n + table[coeff]];
}
// blockIdx.z = polynomial of a batch (stride = gridDim.y limbs)
__global__ __launch_bounds__(256) void galois_coeff_kernel(u64 *dst, const u64 *src, const DModulus *mod,
                                                           uint32_t mod_start, uint32_t elt_inv, uint32_t n) {
    const uint32_t limb = blockIdx.z * gridDim.y + blockIdx.y;
    const u64 q = mod[mod_start + blockIdx.y].value;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    // index_raw = c * galois_elt mod 2n sends coefficient c to position index_raw mod n, negated when index_raw >= n
    // (include/galois.cuh:115-130).  As a gather (contiguous stores): position `coeff` comes from c = coeff * elt^-1 mod 2n, from
    // c - n and negated when c >= n; `elt_inv` = elt^-1 mod 2n (inv_mod_2n below)
    const uint32_t c = (uint32_t)(((u64)coeff * elt_inv) & (2 * (u64)n - 1));
    u64 v = src[(size_t)limb * n + (c & (n - 1))];
    if (c >= n) v = neg_mod(v, q);
    dst[(size_t)limb * n + coeff] = v;
}

// Galois automorphism of a batch of size-2 ciphertexts laid out for the key switch that follows it (rotate_internal /
// apply_galois_inplace, src/evaluate.cu:1567-1624): polynomial 0 goes to dst_ct[b][0], dst_ct[b][1] is zeroed and polynomial 1
// goes to the dense key-switch operand dst_c2[b] -- one kernel instead of permutation + memset + two strided copies.
// blockIdx.z = 2 b + p.  `table` != null: NTT-domain gather; null: coefficient-domain scatter with sign (galois.cu:11-39).
// (`elt` is the element for the table form and its inverse mod 2n for the coefficient form: see the gather below.)
// `add` != null: the automorphism is applied to src + add (mod q), add being polynomial p of ciphertext b at add + (b * add_ct_polys
// + p) * Ql * N -- the sum ct + keyswitch(c2) of a relinearization that was never stored (pha_relinearize_rotate_batched).
__global__ __launch_bounds__(256) void galois_split_kernel(u64 *dst_ct, u64 *dst_c2, const u64 *src, const uint32_t *table,
                                                           const DModulus *mod, uint32_t elt, uint32_t n, uint32_t ql,
                                                           const u64 *add, uint32_t add_ct_polys) {
    const uint32_t limb = blockIdx.y, b = blockIdx.z >> 1, p = blockIdx.z & 1;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    const u64 *in = src + ((size_t)(2 * b + p) * ql + limb) * n;
    const u64 *in2 = add ? add + ((size_t)(add_ct_polys * b + p) * ql + limb) * n : nullptr;
    u64 *out = p ? dst_c2 + ((size_t)b * ql + limb) * n : dst_ct + ((size_t)(2 * b) * ql + limb) * n;
    if (!p) dst_ct[((size_t)(2 * b + 1) * ql + limb) * n + coeff] = 0;
    const u64 q = mod[limb].value;
    if (table) {
        const uint32_t from = table[coeff];
        u64 v = in[from];
        if (in2) v = add_mod(v, in2[from], q);
        out[coeff] = v;
    } else {
        // X -> X^elt sends coefficient c to position c elt mod n with the sign of (c elt mod 2n) >= n (galois.cu:11-39).  Written
        // as a gather: output `coeff` comes from c = coeff elt^-1 mod 2n (c >= n: from c - n, negated), so the stores of a
        // wavefront are one contiguous run and the scattered side is the loads, which the limb's 256-512 KB in L2 absorb;
        // `elt` holds elt^-1 mod 2n here (the launcher inverts it)
        const uint32_t c = (uint32_t)(((u64)coeff * elt) & (2 * (u64)n - 1));
        const uint32_t from = c & (n - 1);
        u64 v = in[from];
        if (in2) v = add_mod(v, in2[from], q);
        if (c >= n) v = neg_mod(v, q);
        out[coeff] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
static bool ntt_domain_scheme(int scheme) {
    if (scheme == PHA_SCHEME_CKKS || scheme == PHA_SCHEME_BGV) return true;
    if (scheme == PHA_SCHEME_BFV) return false;
    throw std::invalid_argument("unsupported scheme");
}

// DRNSTool::modup rns_bconv.cu:530-627.  All beta digits go through ONE base-conversion launch and ONE
// forward-NTT launch pair (blockIdx.z = digit; digit z skips its own limbs, ntt_modup.cu:422).
// `batch` ciphertexts at once: cks / t_cks are [batch][Ql][N], dst is [batch][beta][QlP][N].
// cks_stride: elements between the c2 polynomials of consecutive ciphertexts (0 = dense, Ql * N)
// fused_ip != null (one ciphertext): the forward transform's contiguous pass carries the key inner product
// (modup_ntt_inner_prod): cx is produced here, the digits' own limbs are neither copied nor transformed, and the function
// returns true; on false (shape without a fused form) nothing but the conversion has happened... the caller must not rely on
// that: it passes fused_ip only when fusable_ip() says so.
static bool fusable_ip(Context &c, Tool &t) { return t.alpha > 1 && c.log_n >= 14 && c.log_n <= 17 && t.beta <= 4; }
// r06: batched key switches of the NTT-form schemes leave the digits' own limbs where they are (modup's own_in_place + InnerArgs::own):
// the batched inner product kernel is the one that knows how to read them from c2
static bool own_in_place_ok(const Tool &t, int scheme, uint32_t batch) {
    return (scheme == PHA_SCHEME_CKKS || scheme == PHA_SCHEME_BGV) && batch > 1 && t.alpha > 1 && t.beta <= 4;
}
// own_in_place (r06; NTT-form schemes, batched key switches): the digits' own limbs are NOT copied into dst -- the caller's inner product
// reads them from cks (InnerArgs::own); dst's own-limb rows stay unwritten and nothing reads them (the forward transform skips them)
static void modup(Context &c, Tool &t, u64 *dst, const u64 *cks, int scheme, u64 *t_cks, hipStream_t s,
                  uint32_t batch = 1, size_t cks_stride = 0, const ModupIpArgs *fused_ip = nullptr, bool own_in_place = false) {
    const size_t n = c.n;
    const uint32_t ql = t.size_ql, qlp = t.size_qlp, alpha = t.alpha;
    const bool ntt_dom = ntt_domain_scheme(scheme);
    if (!cks_stride) cks_stride = (size_t)ql * n;
    if (ntt_dom) {
        NttExtra x;
        x.batch = batch;
        x.poly_stride = (size_t)ql * n;
        x.in_stride = cks_stride;
        if (alpha == 1) {
            ntt_inverse(c, cks, t_cks, t_cks, plain_sel(0, ql), EPI_INV_CANON, x, s);
        } else {
            // iNTT fused with x partQlHatInv (bconv phase 1), :558-559
            x.scale = t.part_hat_inv.p;
            x.scale_shoup = t.part_hat_inv_shoup.p;
            ntt_inverse(c, cks, t_cks, t_cks, plain_sel(0, ql), EPI_INV_SCALE, x, s);
        }
    }
    if (alpha == 1) {
        for (uint32_t g = 0; g < batch; g++)
            for (uint32_t b = 0; b < t.beta; b++) {
                const size_t off = (size_t)g * ql * n + (size_t)b * n, off_c = (size_t)g * cks_stride + (size_t)b * n;
                u64 *out = dst + ((size_t)g * t.beta + b) * qlp * n;
                SinglePArgs k{out, cks + off_c, ntt_dom ? t_cks + off : cks + off_c, c.d_mod.p, t.d_qlp_prime.p, b, (uint32_t)n};
                hipLaunchKernelGGL(single_p_kernel, dim3((unsigned)(n / 256), qlp), dim3(256), 0, s, k);
                check_launch();
            }
    }
    LimbSel sel = special_sel(0, qlp, c.size_qp, c.size_p);
    NttExtra x;
    x.batch = t.beta * batch;
    x.poly_stride = (size_t)qlp * n;
    if (ntt_dom) {  // digit z % beta skips [d*alpha, min((d+1)*alpha, ql))
        sel.excl_start = 0;
        sel.excl_end = alpha;
        x.excl_step = alpha;
        x.excl_limit = ql;
        x.excl_mod = batch > 1 ? t.beta : 0;
    }
    if (alpha > 1) {
        // own limbs are copied verbatim by the same kernel (modup_copy_partQl_kernel :522-528);
        // BFV still needs the q-hat^-1 scaling (bconv_mult_kernel :603-607), ckks/bgv got it in the iNTT
        uint32_t max_osz = 0;
        bool mont_split = t.modup_split == 1;
        for (const BConv &b : t.digit) {
            max_osz = b.osz > max_osz ? b.osz : max_osz;
            mont_split = mont_split && b.split_kind == 1 && b.mont && b.row_pad == (uint32_t)kBcRowPad;
        }
        // r05: NTT-form input, N = 2^16, the separate (batched) inner product: the conversion is the load of the forward
        // transform's strided pass (modup_conv_strided, pha_ntt.hip); the transform then runs its contiguous pass only
        if (ntt_dom && !fused_ip) {
            const ModupConvArgs mc{t.d_digit_convs.p, t.beta, t_cks, (size_t)ql * n, own_in_place ? nullptr : cks, cks_stride, alpha, max_osz, mont_split};
            if (modup_conv_strided(c, dst, sel, x, mc, s)) {
                x.first_pass_done = true;
                ntt_forward(c, dst, dst, dst, sel, EPI_FWD_CANON, x, s);
                return;
            }
        }
        // (coefficient-form input: the conversion reads c2 itself, at its own stride; NTT form: the dense t_cks, and c2 only
        //  for the verbatim copy of the digit's own limbs)
        // (fused inner product, NTT-form input: the own limbs are read from c2 by the fused pass, no copy)
        launch_bconv(c, t.d_digit_convs.p, 1, t.beta * batch, alpha, max_osz, t.modup_split, dst, (size_t)qlp * n,
                     ntt_dom ? t_cks : cks, 0, ((fused_ip || own_in_place) && ntt_dom) ? nullptr : cks, !ntt_dom, s, batch > 1 ? t.beta : 0,
                     ntt_dom ? (size_t)ql * n : cks_stride, nullptr, cks_stride);
    }
    if (fused_ip) {
        ModupIpArgs ip = *fused_ip;
        ip.own = ntt_dom ? cks : nullptr;
        if (!modup_ntt_inner_prod(c, dst, sel, x, t.beta, ip, s)) throw std::logic_error("fused inner product requested for a shape without one");
        return;
    }
    ntt_forward(c, dst, dst, dst, sel, EPI_FWD_CANON, x, s);
}

// P^-1 mod q_limb (bigPInv_mod_q rns.cu:110-123) on the host
static u64 h_invmod_p(Context &c, uint32_t limb) {
    const u64 q = c.primes[limb];
    u64 p = 1;
    for (uint32_t k = 0; k < c.size_p; k++) p = h_mulmod(p, c.primes[c.size_q + k] % q, q);
    return h_invmod(p, q);
}

// mod-up + inner product of ONE ciphertext: the fused form where the shape has one, else the two steps.  fix_ct as in inner_prod.
static void inner_prod(Context &c, Tool &t, u64 *cx, const u64 *t_mod_up, const u64 *const *rlk, hipStream_t s, uint32_t batch = 1,
                       const u64 *fix_ct = nullptr, const u64 *own = nullptr, size_t own_stride = 0);
// fold_inverse (ckks forms): the fused kernel also runs the contiguous pass of the mod-down's inverse transform on the special limbs
// (and on the last data limb when fix_ct is given, the fused rescale); returns true when it did -- the caller then launches that
// inverse with NttExtra::second_pass_only.
static bool modup_inner_prod(Context &c, Tool &t, u64 *cx, u64 *t_mod_up, const u64 *c2, const u64 *const *rlk, int scheme, u64 *tmp,
                             hipStream_t s, const u64 *fix_ct = nullptr, bool fold_inverse = false) {
    if (!fusable_ip(c, t)) {
        modup(c, t, t_mod_up, c2, scheme, tmp, s);
        inner_prod(c, t, cx, t_mod_up, rlk, s, 1, fix_ct);
        return false;
    }
    ModupIpArgs ip{};
    ip.cx = cx; ip.evks = rlk; ip.qlp_n = (size_t)t.size_qlp * c.n; ip.qp_n = (size_t)c.size_qp * c.n;
    ip.fix_limb = 0xffffffffu;
    if (fix_ct) {
        const u64 pinv_last = h_invmod_p(c, t.size_ql - 1);
        ip.fix_limb = t.size_ql - 1;
        ip.fix_cst = u64x2{pinv_last, h_shoup(pinv_last, c.primes[t.size_ql - 1])};
        ip.fix_ct = fix_ct;
        ip.fix_ct_stride = (size_t)t.size_ql * c.n;
    }
    if (fold_inverse) {
        ip.inv_from = t.size_ql;
        ip.inv_lead = fix_ct ? t.size_ql - 1 : 0xffffffffu;
        ip.itw = c.d_itw.p;
        ip.itwf = c.d_itwf.p;
    }
    modup(c, t, t_mod_up, c2, scheme, tmp, s, 1, 0, &ip);
    return ip.inv_from != 0xffffffffu;
}

// phantom::key_switch_inner_prod eval_key_switch.cu:71-92
// fix_ct != null (pha_keyswitch_rescale): cx's last data limb receives ct_last + cx_last * P^-1 (see InnerArgs)
static void inner_prod(Context &c, Tool &t, u64 *cx, const u64 *t_mod_up, const u64 *const *rlk, hipStream_t s,
                       uint32_t batch, const u64 *fix_ct, const u64 *own, size_t own_stride) {
    InnerArgs k{};
    k.cx = cx; k.t_mod_up = t_mod_up; k.evks = rlk; k.mod = c.d_mod.p; k.qlp_prime = t.d_qlp_prime.p;
    k.n = (uint32_t)c.n; k.beta = t.beta; k.qlp_n = (size_t)t.size_qlp * c.n; k.qp_n = (size_t)c.size_qp * c.n;
    k.fix_limb = 0xffffffffu;
    k.fpinfo = c.d_fpinfo.p;
    if (own && !(batch > 1 && t.beta <= 4)) throw std::logic_error("inner_prod: own limbs in place need the batched kernel");
    k.own = own; k.own_stride = own_stride; k.alpha = t.alpha; k.ql = t.size_ql;
    if (fix_ct) {
        const u64 pinv_last = h_invmod_p(c, t.size_ql - 1);
        k.fix_limb = t.size_ql - 1;
        k.fix_cst = u64x2{pinv_last, h_shoup(pinv_last, c.primes[t.size_ql - 1])};
        k.fix_ct = fix_ct;
        k.fix_ct_stride = (size_t)t.size_ql * c.n;
    }
    const dim3 grid((unsigned)(c.n / 512), t.size_qlp), block(256);
    if (batch > 1 && t.beta <= 4) {  // key limbs stay in registers across the ciphertexts
        switch (t.beta) {
            case 1: hipLaunchKernelGGL(inner_prod_batched_kernel<1>, grid, block, 0, s, k, batch); break;
            case 2: hipLaunchKernelGGL(inner_prod_batched_kernel<2>, grid, block, 0, s, k, batch); break;
            case 3: hipLaunchKernelGGL(inner_prod_batched_kernel<3>, grid, block, 0, s, k, batch); break;
            default: hipLaunchKernelGGL(inner_prod_batched_kernel<4>, grid, block, 0, s, k, batch); break;
        }
        check_launch();
        return;
    }
    for (uint32_t b = 0; b < batch; b++) {
        hipLaunchKernelGGL(inner_prod_kernel, grid, block, 0, s, k);
        check_launch();
        k.t_mod_up += (size_t)t.beta * k.qlp_n;
        k.cx += 2 * k.qlp_n;
        if (k.fix_ct) k.fix_ct += 2 * k.fix_ct_stride;
    }
}

// DRNSTool::moddown_from_NTT rns_bconv.cu:776-828 for `polys` polynomials cx + z*cx_stride at once.
// accumulate = false: ct_z = result (the reference call).  accumulate = true: ct_z += result, i.e. the
// add_to_ct_kernel of keyswitch_inplace (rns_bconv.cu:763-769) fused into the NTT epilogue.
// folded (ckks): the special limbs of cx already went through the inverse transform's contiguous pass (modup_inner_prod).
// coeff_input (bfv only): DRNSTool::moddown rns_bconv.cu:712-761 -- cx is already in coefficient form, no inverse transform.
static void moddown_from_ntt(Context &c, Tool &t, u64 *ct, size_t ct_stride, u64 *cx, size_t cx_stride,
                             uint32_t polys, int scheme, bool accumulate, u64 *delta, hipStream_t s, bool folded = false,
                             bool coeff_input = false) {
    const size_t n = c.n;
    const uint32_t ql = t.size_ql, qlp = t.size_qlp;
    NttExtra xb;
    xb.batch = polys;
    xb.poly_stride = cx_stride;
    // ckks, alpha > 1: bconv phase 1 (x phat_i^-1 mod p_i, bconv_mult_kernel rns_bconv.cu:22-33) rides on the
    // inverse NTT's last round, exactly as mod-up does (:558-559); the conversion then skips its own scaling
    const bool prescaled = scheme == PHA_SCHEME_CKKS && t.alpha > 1;
    if (folded && scheme != PHA_SCHEME_CKKS) throw std::logic_error("folded inverse pass outside the ckks mod-down");
    xb.second_pass_only = folded;
    if (prescaled) {
        xb.scale = t.p_hat_inv_by_limb.p;
        xb.scale_shoup = t.p_hat_inv_by_limb_shoup.p;
        ntt_inverse(c, cx, cx, cx, special_sel(ql, c.size_p, c.size_qp, c.size_p), EPI_INV_SCALE, xb, s);
    } else if (scheme == PHA_SCHEME_CKKS)
        ntt_inverse(c, cx, cx, cx, special_sel(ql, c.size_p, c.size_qp, c.size_p), EPI_INV_CANON, xb, s);
    else if (scheme == PHA_SCHEME_BFV && coeff_input)
        ;   // (rns_bconv.cu:722-730 transforms ckks and bgv only)
    else if (scheme == PHA_SCHEME_BFV || scheme == PHA_SCHEME_BGV)
        ntt_inverse(c, cx, cx, cx, special_sel(0, qlp, c.size_qp, c.size_p), EPI_INV_CANON, xb, s);
    else
        throw std::invalid_argument("unsupported scheme");
    if (coeff_input && scheme != PHA_SCHEME_BFV) throw std::logic_error("coefficient-form mod-down input outside bfv");
    if (scheme == PHA_SCHEME_BGV && !t.bgv_ready)
        throw std::invalid_argument("bgv needs a plain modulus (pha_context_set_plain_modulus)");
    const size_t d_stride = (size_t)ql * n;
    bool bfv_epilogue_fused = false, ckks_conv_fused = false;
    // r06, CKKS batches at N = 2^16 with the prescaled inputs: the P -> Ql conversion is the LOAD of the forward transform's strided
    // pass (mod-down form of modup_conv_s1_kernel, pha_ntt.hip): delta is never written in coefficient form
    if (prescaled && polys > 1 && t.p_to_ql.split_kind == 1 && t.p_to_ql.mont && t.p_to_ql.row_pad == (uint32_t)kBcRowPad) {
        NttExtra xf;
        xf.batch = polys;
        xf.poly_stride = d_stride;
        ModupConvArgs mc{t.d_p_to_ql_conv.p, 1, cx, cx_stride, nullptr, 0, t.alpha, ql, true, nullptr};
        mc.moddown = true;
        ckks_conv_fused = modup_conv_strided(c, delta, plain_sel(0, ql), xf, mc, s);
    }
    if (ckks_conv_fused) {
    } else if (t.alpha == 1) {
        for (uint32_t z = 0; z < polys; z++) {
            SinglePArgs k{delta + z * d_stride, nullptr, cx + z * cx_stride + (size_t)ql * n, c.d_mod.p,
                          t.d_qlp_prime.p, ql, (uint32_t)n};
            hipLaunchKernelGGL(single_p_kernel, dim3((unsigned)(n / 256), ql), dim3(256), 0, s, k);
            check_launch();
        }
    } else {
        // BFV: (cx - delta) * P^-1 (+ add_to_ct) rides on the conversion's output loop, delta is never stored -- for the
        // register-resident converter (alpha <= 32); a wider P takes the conversion into delta and the element-wise kernel below
        BConvEpilogue e{cx, ct, t.pinv2.p, cx_stride, ct_stride, accumulate};
        bfv_epilogue_fused = scheme == PHA_SCHEME_BFV && t.alpha <= 32;
        launch_bconv(c, t.d_p_to_ql_conv.p, 0, polys, t.alpha, ql, t.moddown_split, delta, d_stride, cx, cx_stride,
                     nullptr, !prescaled, s, 0, 0, bfv_epilogue_fused ? &e : nullptr);
    }
    if (scheme == PHA_SCHEME_BGV) {
        // [cx_P]_t lands in the first P limb of each polynomial, then the t-corrected division and the NTT
        PToTArgs pk{cx + (size_t)ql * n, cx_stride, t.p_to_ql.hat_inv.p, t.p_to_ql.d_iprime.p, t.p_hat_mod_t.p,
                    c.d_mod.p, t.t_mod, t.alpha, (uint32_t)n};
        hipLaunchKernelGGL(p_to_t_kernel, dim3((unsigned)(n / 256), polys), dim3(256), 0, s, pk);
        check_launch();
        {   // every polynomial in one launch
            BgvDownArgs k{accumulate ? delta : ct, cx, delta, cx + (size_t)ql * n, t.p_mod_q2.p, t.pinv2.p, c.d_mod.p,
                          t.pinv_mod_t, t.t_mod.value, (uint32_t)n};
            k.dst_stride = accumulate ? d_stride : ct_stride;
            k.cx_stride = cx_stride;
            k.delta_stride = d_stride;
            hipLaunchKernelGGL(bgv_moddown_kernel, dim3((unsigned)(n / 256), ql, polys), dim3(256), 0, s, k);
            check_launch();
        }
        NttExtra x;
        x.batch = polys;
        x.poly_stride = accumulate ? d_stride : ct_stride;
        u64 *buf = accumulate ? delta : ct;
        ntt_forward(c, buf, buf, buf, plain_sel(0, ql), EPI_FWD_CANON, x, s);
        if (accumulate)
            for (uint32_t z = 0; z < polys; z++)
                launch_add(c, ct + z * ct_stride, delta + z * d_stride, ct + z * ct_stride, ql, 0, s);
    } else if (scheme == PHA_SCHEME_CKKS) {
        NttExtra x;  // NTT(delta) fused with (cx - .) * P^-1 (ntt_moddown.cu:106-261)
        x.scale = t.pinv.p;
        x.scale_shoup = t.pinv_shoup.p;
        x.aux = cx;
        x.batch = polys;
        x.poly_stride = d_stride;
        x.out_stride = ct_stride;
        x.aux_stride = cx_stride;
        x.first_pass_done = ckks_conv_fused;   // the conversion already ran as the load of this transform's strided pass
        ntt_forward(c, delta, delta, ct, plain_sel(0, ql), accumulate ? EPI_FWD_MODDOWN_ADD : EPI_FWD_MODDOWN, x, s);
    } else if (!bfv_epilogue_fused) {
        // BFV, alpha = 1 or alpha > 32 (moddown_kernel rns_bconv.cu:680-689): every polynomial in ONE launch,
        // ct (+)= (cx - delta) * P^-1 (for 1 < alpha <= 32 this already happened inside the base conversion above)
        SubMulArgs k{ct, cx, delta, t.pinv2.p, c.d_mod.p, (uint32_t)n};
        k.dst_stride = ct_stride;
        k.cx_stride = cx_stride;
        k.delta_stride = d_stride;
        k.accumulate = accumulate ? 1 : 0;
        hipLaunchKernelGGL(sub_mul_kernel<false>, dim3((unsigned)(n / 256), ql, polys), dim3(256), 0, s, k);
        check_launch();
    }
}

// DRNSTool::divide_and_round_q_last_ntt rns.cu:1160-1184 on `polys` polynomials src [polys][Ql][N] -> dst [polys][Ql-1][N]
static void rescale_ntt(Context &c, Tool &t, u64 *src, uint32_t polys, u64 *dst, hipStream_t s) {
    const size_t n = c.n, size_Ql = t.size_ql, nl = size_Ql - 1;
    // all polynomials of the ciphertext in one launch each (blockIdx.z = polynomial)
    NttExtra xi;
    xi.batch = polys;
    xi.poly_stride = size_Ql * n;
    ntt_inverse(c, src, src, src, plain_sel(nl, 1), EPI_INV_CANON, xi, s);  // ci[last] -> coefficients (rns.cu:1171)
    // NTT(ci[last] mod qj) fused with (ci[j] - .) * q_last^-1 (rns.cu:1173-1182): the reduction modulo qj
    // (divide_and_round_reduce_q_last_kernel) happens as the first pass loads ci[last]; dst doubles as the
    // buffer between the two passes
    NttExtra x;
    x.scale = t.inv_q_last.p;
    x.scale_shoup = t.inv_q_last_shoup.p;
    x.aux = src;
    x.batch = polys;
    x.poly_stride = nl * n;
    x.out_stride = nl * n;
    x.aux_stride = size_Ql * n;
    x.pro_src = src + nl * n;
    x.pro_stride = size_Ql * n;
    ntt_forward(c, dst, dst, dst, plain_sel(0, nl), EPI_FWD_MODDOWN, x, s);
}

// Key switch + CKKS rescale in one (build-defined fusion; equals keyswitch_inplace eval_key_switch.cu:95-182 followed by
// divide_and_round_q_last_ntt rns.cu:1160-1184 bit for bit): dst [B][2][Ql-1][N] = rescale(ct + keyswitch(c2)).
// Against the two calls it drops the mod-down's forward transform over 2 x Ql limbs and the rescale's separate last-limb inverse
// (their work rides on one inverse over the P limbs + the last data limb and ONE forward over 2 x (Ql - 1) limbs), and ct is
// never written.  scratch `base` as laid out by the callers.
static bool keyswitch_rescale_fusable(const Tool &t) {
    // bconv_rescale_kernel hard-codes the 16-row padding and the 30 / 30 cuts of converter kind 1 (Montgomery entries)
    return t.alpha > 1 && t.alpha <= (uint32_t)kBcRowPad && t.split_ok && t.p_to_ql_pinv.split_kind == 1 && t.p_to_ql_pinv.mont &&
           t.p_to_ql_pinv.row_pad == (uint32_t)kBcRowPad && t.size_ql >= 2;
}
static void keyswitch_rescale(Context &c, Tool &t, const u64 *ct, const u64 *c2, const u64 *const *rlk, u64 *dst, uint32_t B,
                              u64 *base, hipStream_t s) {
    const size_t n = c.n, ql = t.size_ql, ql_n = ql * n, qlp_n = (size_t)t.size_qlp * n, nl = ql - 1;
    u64 *tmp = base, *t_mod_up = base + B * 2 * ql_n, *cx = t_mod_up + B * (size_t)t.beta * qlp_n;
    bool folded = false;   // the inverse transform's contiguous pass already ran inside the fused mod-up
    if (B == 1) {
        folded = modup_inner_prod(c, t, cx, t_mod_up, c2, rlk, PHA_SCHEME_CKKS, tmp, s, ct, true);   // cx_last <- ct_last + cx_last * P^-1
    } else {
        const bool in_place = own_in_place_ok(t, PHA_SCHEME_CKKS, B);   // the batched inner product reads the digits' own limbs from c2
        modup(c, t, t_mod_up, c2, PHA_SCHEME_CKKS, tmp, s, B, 0, nullptr, in_place);
        inner_prod(c, t, cx, t_mod_up, rlk, s, B, ct, in_place ? c2 : nullptr, ql_n);
    }
    {   // coefficient form of the P limbs (x phat_i^-1, bconv phase 1) and of the last data limb, both polynomials, one launch pair
        NttExtra xb;
        xb.batch = 2 * B;
        xb.poly_stride = qlp_n;
        xb.scale = t.p_hat_inv_by_limb.p;
        xb.scale_shoup = t.p_hat_inv_by_limb_shoup.p;
        xb.second_pass_only = folded;
        ntt_inverse(c, cx, cx, cx, special_sel(nl, c.size_p + 1, c.size_qp, c.size_p), EPI_INV_SCALE, xb, s);
    }
    NttExtra x;   // out_j = (ct_j + cx_j P^-1 - NTT(v)_j) q_last^-1
    x.scale = t.inv_q_last.p;
    x.scale_shoup = t.inv_q_last_shoup.p;
    x.scale2 = t.pinv.p;
    x.scale2_shoup = t.pinv_shoup.p;
    x.aux = cx;
    x.aux2 = ct;
    x.batch = 2 * B;
    x.poly_stride = ql_n;
    x.out_stride = nl * n;
    x.aux_stride = qlp_n;
    x.aux2_stride = ql_n;
    // r06, batches at N = 2^16: the conversion (+ last-limb fold) is the LOAD of that transform's strided pass (modup_conv_s1_kernel's
    // rescale form, pha_ntt.hip): v is never written in coefficient form
    if (B > 1) {
        const ModupConvArgs mc{t.d_p_to_ql_pinv_conv.p, 1, cx, qlp_n, nullptr, 0, t.alpha, (uint32_t)ql, true, cx + nl * n};
        x.first_pass_done = modup_conv_strided(c, tmp, plain_sel(0, nl), x, mc, s);
    }
    if (!x.first_pass_done) {   // v_j = delta_j P^-1 + (c_last mod q_j)
        BConvRescaleLaunch L{};
        L.conv = t.d_p_to_ql_pinv_conv.p; L.dst = tmp; L.cx = cx; L.dst_stride = ql_n; L.cx_stride = qlp_n;
        L.mod = c.d_mod.p; L.n = (uint32_t)n; L.ql = (uint32_t)ql;
        const uint32_t groups = bconv_groups(c, (uint32_t)nl, 2 * (uint32_t)B);
        L.out_per_block = ((uint32_t)nl + groups - 1) / groups;
        const dim3 grid((unsigned)(n / kBcThreads), groups, 2 * B), block(kBcThreads);
        if (t.alpha <= 2) hipLaunchKernelGGL(bconv_rescale_kernel<2>, grid, block, 0, s, L);
        else if (t.alpha <= 4) hipLaunchKernelGGL(bconv_rescale_kernel<4>, grid, block, 0, s, L);
        else if (t.alpha <= 8) hipLaunchKernelGGL(bconv_rescale_kernel<8>, grid, block, 0, s, L);
        else if (t.alpha == 15) hipLaunchKernelGGL(bconv_rescale_kernel<15>, grid, block, 0, s, L);
        else hipLaunchKernelGGL(bconv_rescale_kernel<16>, grid, block, 0, s, L);
        check_launch();
    }
    ntt_forward(c, tmp, tmp, dst, plain_sel(0, nl), EPI_FWD_KSRESCALE, x, s);
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
    PHA_CTX_BEGIN(ctx)
    need(dst); need(src);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    // src is the bare [P][N] block: the converter's src_limb offset (Ql) must not be applied
    launch_bconv(c, t.d_p_to_ql_conv.p, 0, 1, t.alpha, (uint32_t)size_Ql, t.moddown_split, dst, 0,
                 src - (size_t)size_Ql * c.n, 0, nullptr, true, as_stream(stream));
    PHA_API_END
}

int pha_modup(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *cks, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(dst); need(cks);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    strict_operand(c, "modup input", cks, rows_plain(0, size_Ql), 1, 0, as_stream(stream));
    u64 *t_cks = c.scratch(stream, size_Ql * c.n);
    modup(c, t, dst, cks, scheme, t_cks, as_stream(stream));
    PHA_API_END
}

int pha_key_switch_inner_prod(pha_context_t ctx, size_t size_Ql, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                              const uint64_t *const *rlk, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(p_cx); need(p_t_mod_up); need(rlk);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    {
        Tool &ts = c.tool((uint32_t)size_Ql);
        strict_operand(c, "key_switch_inner_prod t_mod_up", p_t_mod_up, rows_qlp(ts.size_ql, c.size_q, c.size_p), ts.beta,
                       (size_t)ts.size_qlp * c.n, as_stream(stream));
        strict_keys(c, "key_switch_inner_prod key", rlk, ts.beta, (uint32_t)size_Ql, as_stream(stream));
    }
    inner_prod(c, c.tool((uint32_t)size_Ql), p_cx, p_t_mod_up, rlk, as_stream(stream));
    PHA_API_END
}

int pha_moddown_from_NTT(pha_context_t ctx, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i, int scheme,
                         void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct_i); need(cx_i);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    strict_operand(c, "mod-down input", cx_i, rows_qlp(t.size_ql, c.size_q, c.size_p), 1, 0, as_stream(stream));
    u64 *delta = c.scratch(stream, size_Ql * c.n);
    moddown_from_ntt(c, t, ct_i, 0, cx_i, 0, 1, scheme, false, delta, as_stream(stream));
    PHA_API_END
}

// DRNSTool::moddown rns_bconv.cu:712-761.  Against moddown_from_NTT (:776-828): bfv input is in coefficient form already; ckks
// and bgv take the same steps (the reference's separate forward transform of delta + moddown_kernel :748-757 stores the same
// canonical words as the fused epilogue; its alpha = 1 route through bConv_BEHZ :733 / :746 stores what the single-prime kernel
// does: y = x for a one-prime base, then x mod q_j).
int pha_moddown(pha_context_t ctx, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct_i); need(cx_i);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    strict_operand(c, "mod-down input", cx_i, rows_qlp(t.size_ql, c.size_q, c.size_p), 1, 0, as_stream(stream));
    u64 *delta = c.scratch(stream, size_Ql * c.n);
    moddown_from_ntt(c, t, ct_i, 0, cx_i, 0, 1, scheme, false, delta, as_stream(stream), false, scheme == PHA_SCHEME_BFV);
    PHA_API_END
}

int pha_keyswitch_inplace(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *rlk, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(c2); need(rlk);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    strict_operand(c, "keyswitch ct", ct, rows_plain(0, size_Ql), 2, size_Ql * c.n, s);
    strict_operand(c, "keyswitch c2", c2, rows_plain(0, size_Ql), 1, 0, s);
    strict_keys(c, "keyswitch key", rlk, t.beta, (uint32_t)size_Ql, s);
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    // scratch: t_cks / delta [2][Ql][N] | t_mod_up [beta][QlP][N] | cx [2][QlP][N]  (eval_key_switch.cu:151,155)
    u64 *base = c.scratch(stream, 2 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n);
    u64 *tmp = base, *t_mod_up = base + 2 * ql_n, *cx = t_mod_up + (size_t)t.beta * qlp_n;
    const bool folded = modup_inner_prod(c, t, cx, t_mod_up, c2, rlk, scheme, tmp, s, nullptr, scheme == PHA_SCHEME_CKKS);
    // both polynomials at once; ct += moddown(cx) with the add fused into the NTT epilogue
    moddown_from_ntt(c, t, ct, ql_n, cx, qlp_n, 2, scheme, true, tmp, s, folded);
    PHA_API_END
}

int pha_keyswitch_inplace_batched(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2, size_t batch,
                                  const uint64_t *const *rlk, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(c2); need(rlk);
    if (batch == 0) return 0;
    if (batch > 1024) throw std::invalid_argument("batch out of range");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    if ((size_t)t.beta * batch > 65535 || 2 * batch > 65535) throw std::invalid_argument("batch out of range");
    hipStream_t s = as_stream(stream);
    const uint32_t B = (uint32_t)batch;
    strict_operand(c, "keyswitch ct", ct, rows_plain(0, size_Ql), 2 * B, size_Ql * c.n, s);
    strict_operand(c, "keyswitch c2", c2, rows_plain(0, size_Ql), B, size_Ql * c.n, s);
    strict_keys(c, "keyswitch key", rlk, t.beta, (uint32_t)size_Ql, s);
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    // scratch: t_cks / delta [B][2][Ql][N] | t_mod_up [B][beta][QlP][N] | cx [B][2][QlP][N]
    u64 *base = c.scratch(stream, B * (2 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n));
    u64 *tmp = base, *t_mod_up = base + B * 2 * ql_n, *cx = t_mod_up + B * (size_t)t.beta * qlp_n;
    // (NTT-form schemes, B > 1: the batched inner product reads the digits' own limbs from c2, the mod-up does not copy them)
    const bool in_place = own_in_place_ok(t, scheme, B);
    modup(c, t, t_mod_up, c2, scheme, tmp, s, B, 0, nullptr, in_place);
    inner_prod(c, t, cx, t_mod_up, rlk, s, B, nullptr, in_place ? c2 : nullptr, ql_n);
    // 2B polynomials: ct [B][2][Ql][N] and cx [B][2][QlP][N] are uniformly strided
    moddown_from_ntt(c, t, ct, ql_n, cx, qlp_n, 2 * B, scheme, true, tmp, s);
    PHA_API_END
}

static bool overlaps(const u64 *a, size_t na, const u64 *b, size_t nb) { return a < b + nb && b < a + na; }

int pha_keyswitch_rescale_batched(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, const uint64_t *c2, size_t batch,
                                  const uint64_t *const *rlk, uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(c2); need(rlk); need(dst);
    if (batch == 0) return 0;
    if (batch > 1024) throw std::invalid_argument("batch out of range");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    if (size_Ql < 2) throw std::invalid_argument("cannot rescale the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    if ((size_t)t.beta * batch > 65535 || 2 * batch > 65535) throw std::invalid_argument("batch out of range");
    hipStream_t s = as_stream(stream);
    const uint32_t B = (uint32_t)batch;
    strict_operand(c, "keyswitch ct", ct, rows_plain(0, size_Ql), 2 * B, size_Ql * c.n, s);
    strict_operand(c, "keyswitch c2", c2, rows_plain(0, size_Ql), B, size_Ql * c.n, s);
    strict_keys(c, "keyswitch key", rlk, t.beta, (uint32_t)size_Ql, s);
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    if (overlaps(dst, B * 2 * (size_Ql - 1) * n, ct, B * 2 * ql_n) || overlaps(dst, B * 2 * (size_Ql - 1) * n, c2, B * ql_n))
        throw std::invalid_argument("dst must not overlap ct or c2");
    // scratch: t_cks / v [B][2][Ql][N] | t_mod_up [B][beta][QlP][N] | cx [B][2][QlP][N] (| ct copy [B][2][Ql][N] for the fallback)
    const size_t words = B * (2 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n);
    if (keyswitch_rescale_fusable(t)) {
        keyswitch_rescale(c, t, ct, c2, rlk, dst, B, c.scratch(stream, words), s);
    } else {   // alpha = 1, wide primes, wide P: the two reference steps on a copy of ct
        u64 *base = c.scratch(stream, words + B * 2 * ql_n);
        u64 *tmp = base, *t_mod_up = base + B * 2 * ql_n, *cx = t_mod_up + B * (size_t)t.beta * qlp_n, *work = base + words;
        PHA_HIP(hipMemcpyAsync(work, ct, B * 2 * ql_n * sizeof(u64), hipMemcpyDeviceToDevice, s));
        const bool in_place = own_in_place_ok(t, PHA_SCHEME_CKKS, B);
        modup(c, t, t_mod_up, c2, PHA_SCHEME_CKKS, tmp, s, B, 0, nullptr, in_place);
        inner_prod(c, t, cx, t_mod_up, rlk, s, B, nullptr, in_place ? c2 : nullptr, ql_n);
        moddown_from_ntt(c, t, work, ql_n, cx, qlp_n, 2 * B, PHA_SCHEME_CKKS, true, tmp, s);
        rescale_ntt(c, t, work, 2 * B, dst, s);
    }
    PHA_API_END
}

int pha_keyswitch_rescale(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *rlk, uint64_t *dst, void *stream) {
    return pha_keyswitch_rescale_batched(ctx, size_Ql, ct, c2, 1, rlk, dst, stream);
}

// BASELINE config 4 as one call (build-defined composition of relinearize_inplace src/evaluate.cu:1028-1077 and
// apply_galois_inplace :1567-1624 over a batch): out [B][2][Ql][N] = rotate_elt(relinearize(ct3 [B][3][Ql][N])).
// Against the host composition of the two batched key switches it makes no copies at all: the first key switch reads c2 in
// place (strided) and stores only keyswitch(c2); the Galois kernel forms ct + keyswitch(c2) as it permutes, writing the
// operands of the second key switch straight into `out`, which that key switch then completes in place.
int pha_relinearize_rotate_batched(pha_context_t ctx, size_t size_Ql, const uint64_t *ct3, size_t batch,
                                   const uint64_t *const *rlk, const uint64_t *const *glk, uint32_t galois_elt, int scheme,
                                   uint64_t *out, size_t chunk, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct3); need(rlk); need(glk); need(out);
    if (batch == 0) return 0;
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    if (!(galois_elt & 1) || galois_elt >= 2 * c.n) throw std::invalid_argument("Galois element is not valid");
    Tool &t = c.tool((uint32_t)size_Ql);
    const bool ntt_dom = ntt_domain_scheme(scheme);
    if (scheme == PHA_SCHEME_BGV && !t.bgv_ready) throw std::invalid_argument("bgv needs a plain modulus (pha_context_set_plain_modulus)");
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    if (overlaps(out, batch * 2 * ql_n, ct3, batch * 3 * ql_n)) throw std::invalid_argument("out must not overlap ct3");
    // ciphertexts per set of launches: the mod-up digits of a set (beta x (l + alpha) limbs each) should stay within the MALL
    // (measured at N = 2^15 / 30 + 15 limbs: 8 per set 233 us per ciphertext, all 64 at once 268, one by one 334).
    // chunk = 0 also spreads the sets over two internal streams, half a set each: the short kernels of one set fill the tails of
    // the other's (same process, alternating: 6045-6086 against 5909-5920 ciphertexts/s; four streams or two full sets: no better).
    const bool auto_chunk = chunk == 0;
    if (chunk == 0) chunk = std::max<size_t>(1, ((size_t)192 << 20) / ((size_t)t.beta * qlp_n * sizeof(u64)));
    chunk = std::min<size_t>(std::min<size_t>(chunk, batch), 1024);
    while ((size_t)t.beta * chunk > 65535 || 2 * chunk > 65535) chunk /= 2;
    const bool two_lanes = auto_chunk && chunk >= 2 && batch >= 2 * chunk;
    if (two_lanes) chunk /= 2;
    const uint32_t *tab = ntt_dom ? c.galois_table(galois_elt) : nullptr;
    const size_t C = chunk;
    const size_t words = C * (2 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n + 3 * ql_n);
    // one set of B ciphertexts from b0 on stream ls, scratch base: tmp [C][2][Ql][N] | t_mod_up [C][beta][QlP][N] | cx [C][2][QlP][N] |
    // ks [C][2][Ql][N] | g1 [C][Ql][N]
    auto one_set = [&](hipStream_t ls, u64 *base, size_t b0) {
        u64 *tmp = base, *t_mod_up = base + C * 2 * ql_n, *cx = t_mod_up + C * (size_t)t.beta * qlp_n, *ks = cx + C * 2 * qlp_n,
            *g1 = ks + C * 2 * ql_n;
        const uint32_t B = (uint32_t)std::min(C, batch - b0);
        const u64 *in = ct3 + b0 * 3 * ql_n;
        u64 *o = out + b0 * 2 * ql_n;
        // relinearize: ks = keyswitch(c2), c2 read where it lies (every third polynomial)
        const bool in_place = own_in_place_ok(t, scheme, B);   // (CKKS / BGV; the BFV mod-up transforms its own limbs in the digit buffers)
        modup(c, t, t_mod_up, in + 2 * ql_n, scheme, tmp, ls, B, 3 * ql_n, nullptr, in_place);
        inner_prod(c, t, cx, t_mod_up, rlk, ls, B, nullptr, in_place ? in + 2 * ql_n : nullptr, 3 * ql_n);
        moddown_from_ntt(c, t, ks, ql_n, cx, qlp_n, 2 * B, scheme, false, tmp, ls);
        // rotate: (galois(c0 + ks0), 0) and galois(c1 + ks1) in the layout of the second key switch (apply_galois_inplace)
        hipLaunchKernelGGL(galois_split_kernel, dim3((unsigned)(n / 256), (unsigned)size_Ql, 2 * B), dim3(256), 0, ls, o, g1, ks, tab,
                           c.d_mod.p, tab ? galois_elt : inv_mod_2n(galois_elt, n), (uint32_t)n, (uint32_t)size_Ql, in, 3u);
        check_launch();
        modup(c, t, t_mod_up, g1, scheme, tmp, ls, B, 0, nullptr, in_place);
        inner_prod(c, t, cx, t_mod_up, glk, ls, B, nullptr, in_place ? g1 : nullptr, ql_n);
        moddown_from_ntt(c, t, o, ql_n, cx, qlp_n, 2 * B, scheme, true, tmp, ls);
    };
    if (!two_lanes) {
        u64 *base = c.scratch(stream, words);
        for (size_t b0 = 0; b0 < batch; b0 += C) one_set(s, base, b0);
    } else {
        std::lock_guard<std::mutex> lk(c.lanes.mu);
        c.lanes.ensure();
        u64 *base[2] = {c.scratch(c.lanes.s[0], words), c.scratch(c.lanes.s[1], words)};
        // (both lane arenas are reserved above, before the fork: nothing below allocates)
        PHA_HIP(hipEventRecord(c.lanes.fork, s));
        for (int l = 0; l < 2; l++) PHA_HIP(hipStreamWaitEvent(c.lanes.s[l], c.lanes.fork, 0));
        auto join = [&]() {   // whatever the lanes have been given so far is ordered before anything the caller enqueues on `stream` next
            for (int l = 0; l < 2; l++)
                if (hipEventRecord(c.lanes.join[l], c.lanes.s[l]) == hipSuccess) (void)hipStreamWaitEvent(s, c.lanes.join[l], 0);
        };
        try {
            size_t set = 0;
            for (size_t b0 = 0; b0 < batch; b0 += C, set++) one_set(c.lanes.s[set & 1], base[set & 1], b0);
        } catch (...) {   // a failed launch must not leave work in flight on the internal streams behind an error code
            join();
            throw;
        }
        for (int l = 0; l < 2; l++) {
            PHA_HIP(hipEventRecord(c.lanes.join[l], c.lanes.s[l]));
            PHA_HIP(hipStreamWaitEvent(s, c.lanes.join[l], 0));
        }
    }
    PHA_API_END
}

int pha_hoisting(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                 const uint64_t *const *const *glk, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(galois_elts); need(glk);
    if (n_elts == 0) throw std::invalid_argument("steps must not be empty");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    if (strict_mode()) {
        strict_operand(c, "hoisting ct", ct, rows_plain(0, size_Ql), 2, size_Ql * c.n, s);
        for (size_t e = 0; e < n_elts; e++)
            if (glk[e]) strict_keys(c, "hoisting Galois key", glk[e], t.beta, (uint32_t)size_Ql, s);
    }
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    const bool ntt_dom = ntt_domain_scheme(scheme);
    // per-element device tables: permutation tables and key pointer tables
    std::vector<const uint32_t *> h_tabs(n_elts);
    for (size_t e = 0; e < n_elts; e++) {
        if (!glk[e]) throw std::logic_error("Galois key not present in hoisting");
        h_tabs[e] = c.galois_table(galois_elts[e]);
    }
    // scratch: c0 copy [Ql][N] | tmp / delta [2][Ql][N] | mod-up [beta][QlP][N] | acc_cx [2][QlP][N] | pointer tables
    const size_t ptr_words = 2 * n_elts;
    u64 *base = c.scratch(stream, 3 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n + ptr_words);
    u64 *c0 = base, *tmp = c0 + ql_n, *t_mod_up = tmp + 2 * ql_n, *acc_cx = t_mod_up + (size_t)t.beta * qlp_n;
    u64 *d_ptrs = acc_cx + 2 * qlp_n;
    PHA_HIP(hipMemcpyAsync(d_ptrs, h_tabs.data(), n_elts * sizeof(void *), hipMemcpyHostToDevice, s));
    PHA_HIP(hipMemcpyAsync(d_ptrs + n_elts, glk, n_elts * sizeof(void *), hipMemcpyHostToDevice, s));
    const uint32_t *const *d_tabs = reinterpret_cast<const uint32_t *const *>(d_ptrs);
    const u64 *const *const *d_keys = reinterpret_cast<const u64 *const *const *>(d_ptrs + n_elts);

    PHA_HIP(hipMemcpyAsync(c0, ct, ql_n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    // one mod-up of c1 shared by every rotation (evaluate.cu:1758-1760)
    modup(c, t, t_mod_up, ct + ql_n, scheme, tmp, s);
    // all rotations' inner products in one kernel; the unreduced 128-bit accumulators hold floor(2^128 / q_max^2) - 1
    // products (255 for primes up to 60 bits, 63 for the 61-bit primes the context also accepts)
    u64 qmax = 0;
    for (uint32_t i = 0; i < c.size_qp; i++) qmax = std::max(qmax, c.primes[i]);
    int qbits = 0;
    while (qbits < 64 && (qmax >> qbits)) qbits++;
    const size_t max_terms = qbits >= 64 ? 1 : ((size_t)1 << std::min(20, 128 - 2 * qbits)) - 1;
    const size_t per_call = std::max<size_t>(1, max_terms / t.beta);
    for (size_t e0 = 0; e0 < n_elts; e0 += per_call) {
        HoistArgs k{};
        k.cx = acc_cx; k.t_mod_up = t_mod_up; k.keys = d_keys + e0; k.tables = d_tabs + e0; k.mod = c.d_mod.p;
        k.qlp_prime = t.d_qlp_prime.p; k.n = (uint32_t)n; k.beta = t.beta;
        k.n_elts = (uint32_t)std::min(per_call, n_elts - e0); k.accumulate = e0 ? 1 : 0;
        k.qlp_n = qlp_n; k.qp_n = (size_t)c.size_qp * n;
        const dim3 hgrid((unsigned)(n / 512), t.size_qlp), hblock(256);
        switch (t.beta) {
            case 1: hipLaunchKernelGGL(hoist_inner_prod_kernel<1>, hgrid, hblock, 0, s, k); break;
            case 2: hipLaunchKernelGGL(hoist_inner_prod_kernel<2>, hgrid, hblock, 0, s, k); break;
            case 3: hipLaunchKernelGGL(hoist_inner_prod_kernel<3>, hgrid, hblock, 0, s, k); break;
            case 4: hipLaunchKernelGGL(hoist_inner_prod_kernel<4>, hgrid, hblock, 0, s, k); break;
            default: hipLaunchKernelGGL(hoist_inner_prod_generic_kernel, hgrid, hblock, 0, s, k); break;
        }
        check_launch();
    }
    // ct0 <- sum_e galois_e(c0) ; ct1 <- 0 ; then both += moddown(acc_cx) (fused into the NTT epilogue)
    if (ntt_dom) {
        hipLaunchKernelGGL(hoist_c0_kernel, dim3((unsigned)(n / 256), (unsigned)size_Ql), dim3(256), 0, s, ct, c0,
                           d_tabs, (uint32_t)n_elts, c.d_mod.p, (uint32_t)n);
        check_launch();
    } else {
        PHA_HIP(hipMemsetAsync(ct, 0, ql_n * sizeof(u64), s));
        for (size_t e = 0; e < n_elts; e++) {  // coefficient-domain automorphism (src/galois.cu:20-39)
            hipLaunchKernelGGL(galois_coeff_kernel, dim3((unsigned)(n / 256), (unsigned)size_Ql), dim3(256), 0, s,
                               tmp, c0, c.d_mod.p, 0u, inv_mod_2n(galois_elts[e], n), (uint32_t)n);
            check_launch();
            launch_add(c, ct, tmp, ct, size_Ql, 0, s);
        }
    }
    PHA_HIP(hipMemsetAsync(ct + ql_n, 0, ql_n * sizeof(u64), s));
    moddown_from_ntt(c, t, ct, ql_n, acc_cx, qlp_n, 2, scheme, true, tmp, s);
    PHA_API_END
}

int pha_hoisting_weighted(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                          const uint64_t *const *const *glk, const uint64_t *const *weights, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(galois_elts); need(glk); need(weights);
    if (n_elts == 0) throw std::invalid_argument("steps must not be empty");
    if (!ntt_domain_scheme(scheme)) throw std::invalid_argument("weighted hoisting takes NTT-form ciphertexts (ckks / bgv)");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    if (strict_mode()) {
        strict_operand(c, "hoisting ct", ct, rows_plain(0, size_Ql), 2, size_Ql * c.n, s);
        for (size_t e = 0; e < n_elts; e++) {
            if (glk[e]) strict_keys(c, "hoisting Galois key", glk[e], t.beta, (uint32_t)size_Ql, s);
            if (weights[e]) strict_operand(c, "hoisting weight", weights[e], rows_qlp(t.size_ql, c.size_q, c.size_p), 1, 0, s);
        }
    }
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    // order the elements: key-switched ones first, main-diagonal ones (element 1, no key) last
    std::vector<const void *> tabs, keys, w_ks, w_all;
    std::vector<size_t> order;
    for (int pass = 0; pass < 2; pass++)
        for (size_t e = 0; e < n_elts; e++) {
            if (!weights[e]) throw std::invalid_argument("null weight");
            if ((galois_elts[e] == 1) == (pass == 1)) order.push_back(e);
        }
    size_t n_ks = 0;
    for (size_t e : order) {
        tabs.push_back(c.galois_table(galois_elts[e]));
        w_all.push_back(weights[e]);
        if (galois_elts[e] != 1) {
            if (!glk[e]) throw std::logic_error("Galois key not present in hoisting");
            keys.push_back(glk[e]);
            n_ks++;
        }
    }
    // scratch: (c0, c1) copy [2][Ql][N] | tmp / delta [2][Ql][N] | mod-up [beta][QlP][N] | acc_cx [2][QlP][N] | pointers
    const size_t ptr_words = 3 * n_elts;
    u64 *base = c.scratch(stream, 4 * ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n + ptr_words);
    u64 *cc = base, *tmp = cc + 2 * ql_n, *t_mod_up = tmp + 2 * ql_n, *acc_cx = t_mod_up + (size_t)t.beta * qlp_n;
    u64 *d_ptrs = acc_cx + 2 * qlp_n;
    PHA_HIP(hipMemcpyAsync(d_ptrs, tabs.data(), n_elts * sizeof(void *), hipMemcpyHostToDevice, s));
    PHA_HIP(hipMemcpyAsync(d_ptrs + n_elts, w_all.data(), n_elts * sizeof(void *), hipMemcpyHostToDevice, s));
    if (n_ks) PHA_HIP(hipMemcpyAsync(d_ptrs + 2 * n_elts, keys.data(), n_ks * sizeof(void *), hipMemcpyHostToDevice, s));
    const uint32_t *const *d_tabs = reinterpret_cast<const uint32_t *const *>(d_ptrs);
    const u64 *const *d_w = reinterpret_cast<const u64 *const *>(d_ptrs + n_elts);
    const u64 *const *const *d_keys = reinterpret_cast<const u64 *const *const *>(d_ptrs + 2 * n_elts);
    PHA_HIP(hipMemcpyAsync(cc, ct, 2 * ql_n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    if (n_ks) {
        modup(c, t, t_mod_up, ct + ql_n, scheme, tmp, s);
        const size_t per_call = 63;    // weighted terms below 2^122 each (61-bit primes) in the 128-bit accumulator
        for (size_t e0 = 0; e0 < n_ks; e0 += per_call) {
            HoistWArgs kw{};
            HoistArgs &k = kw.h;
            k.cx = acc_cx; k.t_mod_up = t_mod_up; k.keys = d_keys + e0; k.tables = d_tabs + e0; k.mod = c.d_mod.p;
            k.qlp_prime = t.d_qlp_prime.p; k.n = (uint32_t)n; k.beta = t.beta;
            k.n_elts = (uint32_t)std::min(per_call, n_ks - e0); k.accumulate = e0 ? 1 : 0;
            k.qlp_n = qlp_n; k.qp_n = (size_t)c.size_qp * n;
            kw.weights = d_w + e0;
            const dim3 hgrid((unsigned)(n / 512), t.size_qlp), hblock(256);
            switch (t.beta) {
                case 1: hipLaunchKernelGGL(hoist_weighted_inner_prod_kernel<1>, hgrid, hblock, 0, s, kw); break;
                case 2: hipLaunchKernelGGL(hoist_weighted_inner_prod_kernel<2>, hgrid, hblock, 0, s, kw); break;
                case 3: hipLaunchKernelGGL(hoist_weighted_inner_prod_kernel<3>, hgrid, hblock, 0, s, kw); break;
                case 4: hipLaunchKernelGGL(hoist_weighted_inner_prod_kernel<4>, hgrid, hblock, 0, s, kw); break;
                default: hipLaunchKernelGGL(hoist_weighted_inner_prod_kernel<0>, hgrid, hblock, 0, s, kw); break;
            }
            check_launch();
        }
    }
    // ct0 <- sum_e w_e galois_e(c0), ct1 <- sum over the main-diagonal elements of w_e c1; then both += moddown(acc_cx)
    hipLaunchKernelGGL(hoist_weighted_c_kernel, dim3((unsigned)(n / 256), (unsigned)size_Ql, 2), dim3(256), 0, s, ct, cc,
                       d_tabs, d_w, (uint32_t)n_elts, (uint32_t)n_ks, c.d_mod.p, (uint32_t)n, ql_n);
    check_launch();
    if (n_ks) moddown_from_ntt(c, t, ct, ql_n, acc_cx, qlp_n, 2, scheme, true, tmp, s);
    PHA_API_END
}

// Core of the baby-step / giant-step entries: `blocks` row blocks that share the input ciphertext and every Galois key.  Block r
// with weights w[r][i][j] gives out[r] = sum_i rot_{G_i}(sum_j w[r][i][j] (.) rot_{B_j}(ct)).  The fused baby-step kernel treats
// the (block, giant step) pairs as one list of accumulators, so the baby keys, the gathered digits and the per-baby reductions are
// paid once for up to 8 of them; everything after it is batched over the blocks.  Every block's result is bit-identical to a
// one-block call.
static void bsgs_core(Context &c, Tool &t, const u64 *ct_in, size_t blocks, const uint32_t *baby_elts, size_t nb,
                      const uint64_t *const *const *baby_glk, const uint32_t *giant_elts, size_t ng,
                      const uint64_t *const *const *giant_glk, const uint64_t *const *weights, u64 *out, int scheme, void *stream) {
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, size_Ql = t.size_ql, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n, G = blocks * ng;
    if (strict_mode()) {
        strict_operand(c, "hoisting ct", ct_in, rows_plain(0, size_Ql), 2, ql_n, s);
        for (size_t j = 0; j < nb; j++)
            if (baby_glk[j]) strict_keys(c, "baby-step Galois key", baby_glk[j], t.beta, (uint32_t)size_Ql, s);
        for (size_t i = 0; i < ng; i++)
            if (giant_glk[i]) strict_keys(c, "giant-step Galois key", giant_glk[i], t.beta, (uint32_t)size_Ql, s);
        for (size_t i = 0; i < G * nb; i++)
            if (weights[i]) strict_operand(c, "BSGS weight", weights[i], rows_qlp(t.size_ql, c.size_q, c.size_p), 1, 0, s);
    }
    // 128-bit accumulators: every weighted term is below 2^122 (61-bit primes), 63 of them fit; the giant inner products are
    // plain products (2^120 each for 60-bit primes)
    size_t nbk = 0, nk = 0;
    for (size_t j = 0; j < nb; j++) {
        if (baby_elts[j] != 1 && !baby_glk[j]) throw std::logic_error("Galois key not present in hoisting");
        nbk += baby_elts[j] != 1;
    }
    for (size_t i = 0; i < ng; i++) {
        if (giant_elts[i] != 1 && !giant_glk[i]) throw std::logic_error("Galois key not present in hoisting");
        nk += giant_elts[i] != 1;
    }
    // what the unreduced 128-bit accumulators of the fused baby-step kernel hold: one s * w product per baby ENTRY (identity entries and
    // repeated elements included) plus the P-scaled c0 / c1 term, each below q_max^2 -- floor(2^128 / q_max^2) - 1 terms (255 for
    // primes up to 60 bits, 63 for the 61-bit primes the context also accepts); the giant steps' key products likewise
    u64 qmax = 0;
    for (uint32_t i = 0; i < c.size_qp; i++) qmax = std::max(qmax, c.primes[i]);
    int qbits = 0;
    while (qbits < 64 && (qmax >> qbits)) qbits++;
    const size_t max_terms = qbits >= 64 ? 1 : ((size_t)1 << std::min(20, 128 - 2 * qbits)) - 1;
    if (nb + 1 > max_terms || nk * t.beta > max_terms || nbk > 255)
        throw std::invalid_argument("too many steps for one call: the 128-bit accumulators hold floor(2^128 / q_max^2) - 1 products (at most "
                                    "254 baby entries for primes up to 60 bits, 62 for 61-bit primes)");
    if (2 * G > 65535 || blocks * nk * t.beta > 65535) throw std::invalid_argument("too many row blocks for one call");
    if (t.beta > 4) throw std::invalid_argument("more than 4 key-switch digits are not supported by the baby-step / giant-step form");
    bool any_null = false;
    for (size_t i = 0; i < G * nb; i++) any_null = any_null || !weights[i];
    std::vector<const void *> h_btab(nb), h_bkeys(nb), h_gtab(ng), h_gkeys;
    std::vector<uint32_t> h_rank(ng);
    for (size_t j = 0; j < nb; j++) {
        h_btab[j] = c.galois_table(baby_elts[j]);
        h_bkeys[j] = baby_elts[j] == 1 ? nullptr : baby_glk[j];
    }
    for (size_t i = 0; i < ng; i++) {
        h_gtab[i] = c.galois_table(giant_elts[i]);
        h_rank[i] = giant_elts[i] == 1 ? 0xffffffffu : (uint32_t)h_gkeys.size();
        if (giant_elts[i] != 1) h_gkeys.push_back(giant_glk[i]);
    }
    // scratch: cc [2][Ql][N] | tmp [max(2 G, blocks nk, 2 blocks)][Ql][N] | mod-up [beta][QlP][N] | acc [G][2][QlP][N] |
    //          B [G][2][Ql][N] | g1 [blocks][nk][Ql][N] | giant mod-up [blocks][nk][beta][QlP][N] | cx [blocks][2][QlP][N] | tables
    const size_t n_tmp = std::max<size_t>(2 * G, std::max<size_t>(blocks * nk, 2 * blocks));
    const size_t ptr_words = 2 * nb + G * nb + 2 * ng + nk + (any_null ? qlp_n : 0);   // (+ a zero plane for the missing weights)
    u64 *base = c.scratch(stream, 2 * ql_n + n_tmp * ql_n + (size_t)t.beta * qlp_n + G * 2 * qlp_n + G * 2 * ql_n + blocks * nk * ql_n +
                                      blocks * nk * (size_t)t.beta * qlp_n + blocks * 2 * qlp_n + ptr_words);
    u64 *cc = base, *tmp = cc + 2 * ql_n, *t_mod_up = tmp + n_tmp * ql_n, *acc = t_mod_up + (size_t)t.beta * qlp_n,
        *B = acc + G * 2 * qlp_n, *g1 = B + G * 2 * ql_n, *mu_g = g1 + blocks * nk * ql_n,
        *cxg = mu_g + blocks * nk * (size_t)t.beta * qlp_n, *d_ptrs = cxg + blocks * 2 * qlp_n;
    u64 *p_btab = d_ptrs, *p_bkeys = p_btab + nb, *p_w = p_bkeys + nb, *p_gtab = p_w + G * nb, *p_rank = p_gtab + ng, *p_gkeys = p_rank + ng;
    PHA_HIP(hipMemcpyAsync(p_btab, h_btab.data(), nb * sizeof(void *), hipMemcpyHostToDevice, s));
    PHA_HIP(hipMemcpyAsync(p_bkeys, h_bkeys.data(), nb * sizeof(void *), hipMemcpyHostToDevice, s));
    std::vector<const void *> h_w(weights, weights + G * nb);
    if (any_null) {
        u64 *zero = p_gkeys + nk;
        PHA_HIP(hipMemsetAsync(zero, 0, qlp_n * sizeof(u64), s));
        for (auto &w : h_w)
            if (!w) w = zero;
    }
    PHA_HIP(hipMemcpyAsync(p_w, h_w.data(), G * nb * sizeof(void *), hipMemcpyHostToDevice, s));
    PHA_HIP(hipMemcpyAsync(p_gtab, h_gtab.data(), ng * sizeof(void *), hipMemcpyHostToDevice, s));
    PHA_HIP(hipMemcpyAsync(p_rank, h_rank.data(), ng * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (nk) PHA_HIP(hipMemcpyAsync(p_gkeys, h_gkeys.data(), nk * sizeof(void *), hipMemcpyHostToDevice, s));
    const uint32_t *const *d_btab = reinterpret_cast<const uint32_t *const *>(p_btab);
    const u64 *const *const *d_bkeys = reinterpret_cast<const u64 *const *const *>(p_bkeys);
    const u64 *const *d_w = reinterpret_cast<const u64 *const *>(p_w);
    const uint32_t *const *d_gtab = reinterpret_cast<const uint32_t *const *>(p_gtab);
    const u64 *const *const *d_gkeys = reinterpret_cast<const u64 *const *const *>(p_gkeys);

    PHA_HIP(hipMemcpyAsync(cc, ct_in, 2 * ql_n * sizeof(u64), hipMemcpyDeviceToDevice, s));   // (out may be ct_in: one-block in-place form)
    // baby steps: one mod-up of c1, then every (block, giant step)'s weighted sum of the hoisted inner products (and of the c0 / c1
    // terms, pre-multiplied by P) in one pass over the baby keys per 8 accumulators; B = moddown(acc), one batched launch set
    if (nbk) modup(c, t, t_mod_up, cc + ql_n, scheme, tmp, s);
    {
        BsgsArgs k{};
        k.acc = acc; k.t_mod_up = t_mod_up; k.keys = d_bkeys; k.tables = d_btab; k.weights = d_w; k.mod = c.d_mod.p;
        k.qlp_prime = t.d_qlp_prime.p; k.n = (uint32_t)n; k.beta = t.beta; k.nb = (uint32_t)nb;
        k.qlp_n = qlp_n; k.qp_n = (size_t)c.size_qp * n;
        k.cc = cc; k.p_mod_q = t.p_mod_q2.p; k.ql = (uint32_t)size_Ql; k.ql_n = ql_n;
        k.fpinfo = c.d_fpinfo.p;
        const dim3 grid((unsigned)(n / 256), t.size_qlp), block(256);
#define PHA_BSGS_GO(NG)                                                                                                   \
    do {                                                                                                                  \
        switch (t.beta) {                                                                                                 \
            case 1: hipLaunchKernelGGL((hoist_bsgs_inner_prod_kernel<NG, 1>), grid, block, 0, s, k); break;              \
            case 2: hipLaunchKernelGGL((hoist_bsgs_inner_prod_kernel<NG, 2>), grid, block, 0, s, k); break;              \
            case 3: hipLaunchKernelGGL((hoist_bsgs_inner_prod_kernel<NG, 3>), grid, block, 0, s, k); break;              \
            default: hipLaunchKernelGGL((hoist_bsgs_inner_prod_kernel<NG, 4>), grid, block, 0, s, k); break;             \
        }                                                                                                                 \
    } while (0)
        for (size_t g0 = 0; g0 < G;) {
            const size_t left = G - g0;
            k.g0 = (uint32_t)g0;
            if (left >= 16) { PHA_BSGS_GO(16); g0 += 16; }
            else if (left >= 8) { PHA_BSGS_GO(8); g0 += 8; }
            else if (left >= 4) { PHA_BSGS_GO(4); g0 += 4; }
            else if (left >= 2) { PHA_BSGS_GO(2); g0 += 2; }
            else { PHA_BSGS_GO(1); g0 += 1; }
            check_launch();
        }
#undef PHA_BSGS_GO
    }
    moddown_from_ntt(c, t, B, ql_n, acc, qlp_n, (uint32_t)(2 * G), scheme, false, tmp, s);
    // giant steps: permutations (every block in one launch), then per block ONE mod-down for the sum of its key-switch inner products
    hipLaunchKernelGGL(bsgs_combine_kernel, dim3((unsigned)(n / 256), (unsigned)size_Ql, (unsigned)blocks), dim3(256), 0, s, out, g1, B,
                       d_gtab, reinterpret_cast<const uint32_t *>(p_rank), (uint32_t)ng, (uint32_t)nk, c.d_mod.p, (uint32_t)n, ql_n);
    check_launch();
    if (nk) {
        modup(c, t, mu_g, g1, scheme, tmp, s, (uint32_t)(blocks * nk));
        MultiInnerArgs k{cxg, mu_g, d_gkeys, c.d_mod.p, t.d_qlp_prime.p, (uint32_t)n, t.beta, (uint32_t)nk, qlp_n, (size_t)c.size_qp * n};
        hipLaunchKernelGGL(inner_prod_multi_kernel, dim3((unsigned)(n / 512), t.size_qlp, (unsigned)blocks), dim3(256), 0, s, k);
        check_launch();
        moddown_from_ntt(c, t, out, ql_n, cxg, qlp_n, (uint32_t)(2 * blocks), scheme, true, tmp, s);
    }
}

int pha_hoisting_weighted_bsgs(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *baby_elts, size_t n_baby,
                               const uint64_t *const *const *baby_glk, const uint32_t *giant_elts, size_t n_giant,
                               const uint64_t *const *const *giant_glk, const uint64_t *const *weights, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(baby_elts); need(baby_glk); need(giant_elts); need(giant_glk); need(weights);
    if (n_baby == 0 || n_giant == 0) throw std::invalid_argument("steps must not be empty");
    if (!ntt_domain_scheme(scheme)) throw std::invalid_argument("weighted hoisting takes NTT-form ciphertexts (ckks / bgv)");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    bsgs_core(c, c.tool((uint32_t)size_Ql), ct, 1, baby_elts, n_baby, baby_glk, giant_elts, n_giant, giant_glk, weights, ct, scheme, stream);
    PHA_API_END
}

int pha_hoisting_weighted_bsgs_blocks(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, size_t n_blocks, const uint32_t *baby_elts,
                                      size_t n_baby, const uint64_t *const *const *baby_glk, const uint32_t *giant_elts, size_t n_giant,
                                      const uint64_t *const *const *giant_glk, const uint64_t *const *weights, uint64_t *out, int scheme,
                                      void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(baby_elts); need(baby_glk); need(giant_elts); need(giant_glk); need(weights); need(out);
    if (n_blocks == 0) return 0;
    if (n_baby == 0 || n_giant == 0) throw std::invalid_argument("steps must not be empty");
    if (!ntt_domain_scheme(scheme)) throw std::invalid_argument("weighted hoisting takes NTT-form ciphertexts (ckks / bgv)");
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    const size_t ql_n = size_Ql * c.n;
    if (overlaps(out, n_blocks * 2 * ql_n, ct, 2 * ql_n)) throw std::invalid_argument("out must not overlap ct");
    bsgs_core(c, c.tool((uint32_t)size_Ql), ct, n_blocks, baby_elts, n_baby, baby_glk, giant_elts, n_giant, giant_glk, weights, out, scheme,
              stream);
    PHA_API_END
}

int pha_divide_and_round_q_last_ntt(pha_context_t ctx, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    Context &c = ctx->c;
    check_level(c, size_Ql, false);
    if (size_Ql < 2) throw std::invalid_argument("cannot rescale the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    if (cipher_size == 0) return 0;
    if (cipher_size > 65535) throw std::invalid_argument("cipher_size out of range");
    rescale_ntt(c, t, src, (uint32_t)cipher_size, dst, as_stream(stream));
    PHA_API_END
}

int pha_generate_one_kswitch_key(pha_context_t ctx, const uint64_t *sk_ntt, const uint64_t *new_key_ntt,
                                 const uint64_t *a, uint64_t *e, uint64_t *const *evk, int scheme, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(sk_ntt); need(new_key_ntt); need(a); need(e); need(evk);
    Context &c = ctx->c;
    if (c.size_p == 0) throw std::invalid_argument("context has no special modulus");
    if (c.size_q % c.size_p) throw std::invalid_argument("#Q must be a multiple of special_modulus_size");
    ntt_domain_scheme(scheme);  // validates the scheme
    Tool &t = c.tool(c.size_q);
    hipStream_t s = as_stream(stream);
    const uint32_t dnum = c.size_q / c.size_p, n = (uint32_t)c.n;
    const size_t qp_n = (size_t)c.size_qp * n;
    if (scheme == PHA_SCHEME_BGV) {  // noise = t*e (secretkey.cu:268-273)
        if (!t.bgv_ready) throw std::invalid_argument("bgv needs a plain modulus (pha_context_set_plain_modulus)");
        hipLaunchKernelGGL(scale_by_t_kernel, dim3(n / 256, c.size_qp, dnum), dim3(256), 0, s, e, c.d_mod.p,
                           t.t_mod.value, n, qp_n);
        check_launch();
    }
    NttExtra x;  // e -> NTT form, every digit in one launch (secretkey.cu:275)
    x.batch = dnum;
    x.poly_stride = qp_n;
    ntt_forward(c, e, e, e, plain_sel(0, c.size_qp), EPI_FWD_CANON, x, s);
    KeyGenArgs k{evk, sk_ntt, new_key_ntt, a, e, t.p_mod_q2.p, c.d_mod.p, n, c.size_qp, c.size_p};
    hipLaunchKernelGGL(kswitch_key_kernel, dim3(n / 256, c.size_qp, dnum), dim3(256), 0, s, k);
    check_launch();
    PHA_API_END
}

int pha_mod_t_and_divide_q_last_ntt(pha_context_t ctx, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    Context &c = ctx->c;
    check_level(c, size_Ql, false);
    if (size_Ql < 2) throw std::invalid_argument("cannot switch down the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    if (!t.bgv_ready) throw std::invalid_argument("bgv needs a plain modulus (pha_context_set_plain_modulus)");
    if (cipher_size == 0) return 0;
    if (cipher_size > 65535) throw std::invalid_argument("cipher_size out of range");
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, nl = size_Ql - 1;
    NttExtra xi;
    xi.batch = (uint32_t)cipher_size;
    xi.poly_stride = size_Ql * n;
    ntt_inverse(c, src, src, src, plain_sel(0, size_Ql), EPI_INV_CANON, xi, s);  // rns.cu:1221
    BgvSwitchArgs k{dst, src, t.q_last_mod_q2.p, t.inv_q_last2.p, c.d_mod.p, t.inv_q_last_mod_t, t.t_mod,
                    (uint32_t)n, (uint32_t)nl, size_Ql * n, nl * n};
    hipLaunchKernelGGL(bgv_switch_kernel, dim3((unsigned)(n / 256), (unsigned)nl, (unsigned)cipher_size), dim3(256), 0,
                       s, k);
    check_launch();
    NttExtra xo;
    xo.batch = (uint32_t)cipher_size;
    xo.poly_stride = nl * n;
    ntt_forward(c, dst, dst, dst, plain_sel(0, nl), EPI_FWD_CANON, xo, s);  // rns.cu:1234
    PHA_API_END
}

int pha_divide_and_round_q_last(pha_context_t ctx, size_t size_Ql, const uint64_t *src, size_t cipher_size,
                                uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    Context &c = ctx->c;
    check_level(c, size_Ql, false);
    if (size_Ql < 2) throw std::invalid_argument("cannot switch down the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    const size_t n = c.n, nl = size_Ql - 1;
    if (cipher_size == 0) return 0;
    if (cipher_size > 65535) throw std::invalid_argument("cipher_size out of range");
    SubMulArgs k{dst, src, src + nl * n, t.inv_q_last2.p, c.d_mod.p, (uint32_t)n};   // every polynomial in one launch
    k.dst_stride = nl * n;
    k.cx_stride = size_Ql * n;
    k.delta_stride = size_Ql * n;
    hipLaunchKernelGGL(sub_mul_kernel<true>, dim3((unsigned)(n / 256), (unsigned)nl, (unsigned)cipher_size), dim3(256), 0,
                       as_stream(stream), k);
    check_launch();
    PHA_API_END
}

int pha_apply_galois_ntt(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t cms,
                         void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    if (src == dst) throw std::invalid_argument("apply_galois_ntt cannot run in place");
    Context &c = ctx->c;
    const uint32_t *tab = c.galois_table(galois_elt);
    hipLaunchKernelGGL(galois_ntt_kernel, dim3((unsigned)(c.n / 256), (unsigned)cms), dim3(256), 0,
                       as_stream(stream), dst, src, tab, (uint32_t)c.n);
    check_launch();
    PHA_API_END
}

int pha_apply_galois_batched(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t cms,
                             size_t polys, int ntt_form, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    if (src == dst) throw std::invalid_argument("apply_galois cannot run in place");
    Context &c = ctx->c;
    if (!(galois_elt & 1) || galois_elt >= 2 * c.n) throw std::invalid_argument("Galois element is not valid");
    if (cms > c.size_qp) throw std::invalid_argument("modulus index out of range");
    if (polys == 0 || cms == 0) return 0;
    if (polys > 65535) throw std::invalid_argument("batch out of range");
    if (ntt_form) {   // the NTT-domain permutation does not depend on the modulus: the polynomials are just more limbs
        const uint32_t *tab = c.galois_table(galois_elt);
        for (size_t p0 = 0; p0 < polys * cms; p0 += 65535) {
            const size_t cnt = std::min<size_t>(65535, polys * cms - p0);
            hipLaunchKernelGGL(galois_ntt_kernel, dim3((unsigned)(c.n / 256), (unsigned)cnt), dim3(256), 0, as_stream(stream),
                               dst + p0 * c.n, src + p0 * c.n, tab, (uint32_t)c.n);
        }
    } else {
        hipLaunchKernelGGL(galois_coeff_kernel, dim3((unsigned)(c.n / 256), (unsigned)cms, (unsigned)polys), dim3(256), 0,
                           as_stream(stream), dst, src, c.d_mod.p, 0u, inv_mod_2n(galois_elt, c.n), (uint32_t)c.n);
    }
    check_launch();
    PHA_API_END
}

int pha_apply_galois_for_keyswitch(pha_context_t ctx, const uint64_t *src, uint64_t *dst_ct, uint64_t *dst_c2,
                                   uint32_t galois_elt, size_t size_Ql, size_t batch, int ntt_form, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst_ct); need(dst_c2);
    if (src == dst_ct || src == dst_c2) throw std::invalid_argument("apply_galois cannot run in place");
    Context &c = ctx->c;
    if (!(galois_elt & 1) || galois_elt >= 2 * c.n) throw std::invalid_argument("Galois element is not valid");
    if (size_Ql == 0 || size_Ql > c.size_q) throw std::invalid_argument("size_Ql out of range");
    if (batch == 0) return 0;
    if (2 * batch > 65535) throw std::invalid_argument("batch out of range");
    const uint32_t *tab = ntt_form ? c.galois_table(galois_elt) : nullptr;
    hipLaunchKernelGGL(galois_split_kernel, dim3((unsigned)(c.n / 256), (unsigned)size_Ql, (unsigned)(2 * batch)), dim3(256), 0,
                       as_stream(stream), dst_ct, dst_c2, src, tab, c.d_mod.p, tab ? galois_elt : inv_mod_2n(galois_elt, c.n), (uint32_t)c.n,
                       (uint32_t)size_Ql, (const u64 *)nullptr, 0u);
    check_launch();
    PHA_API_END
}

int pha_apply_galois(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t cms,
                     size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(src); need(dst);
    if (src == dst) throw std::invalid_argument("apply_galois cannot run in place");
    Context &c = ctx->c;
    if (!(galois_elt & 1) || galois_elt >= 2 * c.n) throw std::invalid_argument("Galois element is not valid");
    if (mod_start + cms > c.size_qp) throw std::invalid_argument("modulus index out of range");
    hipLaunchKernelGGL(galois_coeff_kernel, dim3((unsigned)(c.n / 256), (unsigned)cms), dim3(256), 0,
                       as_stream(stream), dst, src, c.d_mod.p, (uint32_t)mod_start, inv_mod_2n(galois_elt, c.n), (uint32_t)c.n);
    check_launch();
    PHA_API_END
}

}  // extern "C"
