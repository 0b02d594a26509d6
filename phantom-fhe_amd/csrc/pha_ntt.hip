// pha_ntt.hip -- NTT kernels and launchers (gfx950).
//
// One templated kernel per (pass configuration, direction, epilogue); the thread program is
// pha_ntt_core.h.  The 17 launchers of include/ntt.cuh:157-226 that the hot path uses collapse to
// two drivers (forward / inverse) parameterised by a limb selector (start, special-prime remap,
// excluded range: src/ntt/fntt_2d.cu:434-437, src/ntt/ntt_modup.cu:422) and an epilogue
// (canonicalise / fused mod-down / scale), instead of 25 hand-copied kernels.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

namespace pha {

struct NttKArgs {
    const u64 *in;
    u64 *out;
    u64 *mid;                // buffer between the two passes (pass 1: in -> mid, pass 2: mid -> out)
    const u64x2 *tw;         // table base [prime][n] (forward or inverse)
    const DModulus *mod;     // [prime]
    const u64x2 *ninv;       // [prime]
    const u64x2 *w1ninv;     // [prime]
    const u64 *scale;        // [limb] or null
    const u64 *scale_shoup;  // [limb] or null
    const u64 *aux;          // fuse_moddown: cx base
    LimbSel sel;
    uint32_t log_n;
    uint32_t t1, t2;         // N = t1 * t2
};

template <class C, bool FWD, int EPI, bool FOLD>
__global__ __launch_bounds__(kThreads) void ntt_pass_kernel(const NttKArgs k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);

    const uint32_t twr = k.sel.start + blockIdx.y;  // limb in the buffer (uniform)
    if (twr >= k.sel.excl_start && twr < k.sel.excl_end) return;
    const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
    const size_t n = (size_t)1 << k.log_n;

    PassArgs a;
    a.in = k.in + (size_t)twr * n;
    a.out = k.out + (size_t)twr * n;
    a.tw = k.tw + (size_t)prime * n;
    a.q = k.mod[prime].value;
    a.tile = blockIdx.x;
    a.rho0 = k.t1;
    a.stride = k.t2;
    if (!FWD && FOLD) {
        a.ninv = k.ninv[prime];
        a.w1ninv = k.w1ninv[prime];
    }
    if (EPI == EPI_INV_SCALE || EPI == EPI_FWD_MODDOWN) {
        a.scale.x = k.scale[twr];
        a.scale.y = k.scale_shoup[twr];
    }
    a.aux = (EPI == EPI_FWD_MODDOWN) ? k.aux + (size_t)twr * n : nullptr;

    u64 reg[kElemsPerThread];
    using Prog = PassProgram<C, FWD, EPI, FOLD>;
    const int tid = threadIdx.x;
    Prog::template run<0>(a, lds, tid, reg);
    __syncthreads();
    Prog::template run<1>(a, lds, tid, reg);
    if constexpr (Prog::NSEG == 3) {
        __syncthreads();
        Prog::template run<2>(a, lds, tid, reg);
    }
}

template <class C, bool FWD, int EPI, bool FOLD>
static void launch_pass(const NttKArgs &k, hipStream_t s) {
    const size_t n = (size_t)1 << k.log_n;
    dim3 grid((unsigned)(n / kTileElems), k.sel.count);
    const size_t lds_bytes = (size_t)C::LDS_WORDS * sizeof(u64);
    hipLaunchKernelGGL((ntt_pass_kernel<C, FWD, EPI, FOLD>), grid, dim3(kThreads), lds_bytes, s, k);
    check_launch();
}

template <int LOGN>
static void forward_impl(NttKArgs k, int epi, hipStream_t s) {
    using P1 = typename NttPlan<LOGN>::P1;
    using P2 = typename NttPlan<LOGN>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    u64 *const final_out = k.out;
    k.out = k.mid;
    launch_pass<P1, true, EPI_NONE, false>(k, s);
    k.in = k.mid;
    k.out = final_out;
    if (epi == EPI_FWD_MODDOWN) launch_pass<P2, true, EPI_FWD_MODDOWN, false>(k, s);
    else launch_pass<P2, true, EPI_FWD_CANON, false>(k, s);
}

template <int LOGN>
static void inverse_impl(NttKArgs k, int epi, hipStream_t s) {
    using P1 = typename NttPlan<LOGN>::P1;
    using P2 = typename NttPlan<LOGN>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    u64 *const final_out = k.out;
    k.out = k.mid;
    launch_pass<P2, false, EPI_NONE, false>(k, s);
    k.in = k.mid;
    k.out = final_out;
    if (epi == EPI_INV_SCALE) launch_pass<P1, false, EPI_INV_SCALE, true>(k, s);
    else launch_pass<P1, false, EPI_INV_CANON, true>(k, s);
}

static NttKArgs make_args(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, const NttExtra &x,
                          bool fwd) {
    NttKArgs k{};
    k.in = in;
    k.mid = mid;
    k.out = out;
    k.tw = fwd ? c.d_tw.p : c.d_itw.p;
    k.mod = c.d_mod.p;
    k.ninv = c.d_ninv.p;
    k.w1ninv = c.d_w1ninv.p;
    k.scale = x.scale;
    k.scale_shoup = x.scale_shoup;
    k.aux = x.aux;
    k.sel = sel;
    k.log_n = c.log_n;
    return k;
}

static void check_sel(Context &c, const LimbSel &sel) {
    if (sel.count == 0) return;
    const uint32_t last = sel.start + sel.count - 1;
    const uint32_t prime_last = last >= sel.remap_from ? last + sel.remap_add : last;
    if (prime_last >= c.size_qp) throw std::invalid_argument("modulus index out of range of the NTT tables");
}

void ntt_forward(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, true);
    switch (c.log_n) {
        case 12: forward_impl<12>(k, epi, s); break;
        case 13: forward_impl<13>(k, epi, s); break;
        case 14: forward_impl<14>(k, epi, s); break;
        case 15: forward_impl<15>(k, epi, s); break;
        case 16: forward_impl<16>(k, epi, s); break;
        case 17: forward_impl<17>(k, epi, s); break;
        default: throw std::invalid_argument("unsupported polynomial degree");
    }
}

void ntt_inverse(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, false);
    switch (c.log_n) {
        case 12: inverse_impl<12>(k, epi, s); break;
        case 13: inverse_impl<13>(k, epi, s); break;
        case 14: inverse_impl<14>(k, epi, s); break;
        case 15: inverse_impl<15>(k, epi, s); break;
        case 16: inverse_impl<16>(k, epi, s); break;
        case 17: inverse_impl<17>(k, epi, s); break;
        default: throw std::invalid_argument("unsupported polynomial degree");
    }
}

}  // namespace pha

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

extern "C" {

int pha_nwt_2d_radix8_forward_inplace(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start, void *stream) {
    PHA_API_BEGIN
    need(inout);
    ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                          size_t start, size_t size_QP, size_t size_P,
                                                          void *stream) {
    PHA_API_BEGIN
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    ntt_forward(ctx->c, inout, inout, inout, special_sel(start, cms, size_QP, size_P), EPI_FWD_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(pha_context_t ctx, uint64_t *inout,
                                                                        size_t cms, size_t start, size_t size_QP,
                                                                        size_t size_P, size_t ex_start,
                                                                        size_t ex_end, void *stream) {
    PHA_API_BEGIN
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    LimbSel sel = special_sel(start, cms, size_QP, size_P);
    sel.excl_start = (uint32_t)ex_start;
    sel.excl_end = (uint32_t)ex_end;
    ntt_forward(ctx->c, inout, inout, inout, sel, EPI_FWD_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_fuse_moddown(pha_context_t ctx, uint64_t *ct, const uint64_t *cx,
                                                   const uint64_t *pinv, const uint64_t *pinv_shoup,
                                                   uint64_t *delta, size_t cms, size_t start, void *stream) {
    PHA_API_BEGIN
    need(ct); need(cx); need(pinv); need(pinv_shoup); need(delta);
    NttExtra x;
    x.scale = pinv;
    x.scale_shoup = pinv_shoup;
    x.aux = cx;
    // pass 1 in place on delta; pass 2 reads delta, fuses (cx - NTT(delta)) * PInv and writes ct
    // (ntt_moddown.cu:106-261).  ct may alias cx: every thread reads cx[i] before it writes ct[i].
    ntt_forward(ctx->c, delta, delta, ct, plain_sel(start, cms), EPI_FWD_MODDOWN, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start, void *stream) {
    PHA_API_BEGIN
    need(inout);
    ntt_inverse(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms, size_t start,
                               void *stream) {
    PHA_API_BEGIN
    need(out); need(in);
    ntt_inverse(ctx->c, in, out, out, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_scale(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms,
                                     size_t start, const uint64_t *scale, const uint64_t *scale_shoup,
                                     void *stream) {
    PHA_API_BEGIN
    need(out); need(in); need(scale); need(scale_shoup);
    NttExtra x;
    x.scale = scale;
    x.scale_shoup = scale_shoup;
    ntt_inverse(ctx->c, in, out, out, plain_sel(start, cms), EPI_INV_SCALE, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_scale(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                             const uint64_t *scale, const uint64_t *scale_shoup, void *stream) {
    return pha_nwt_2d_radix8_backward_scale(ctx, inout, inout, cms, start, scale, scale_shoup, stream);
}

int pha_nwt_2d_radix8_backward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                           size_t start, size_t size_QP, size_t size_P,
                                                           void *stream) {
    PHA_API_BEGIN
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    ntt_inverse(ctx->c, inout, inout, inout, special_sel(start, cms, size_QP, size_P), EPI_INV_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

int pha_time_forward_ntt(pha_context_t ctx, uint64_t *inout, size_t cms, int iters, void *stream, float *ms_out) {
    PHA_API_BEGIN
    need(inout);
    hipStream_t s = as_stream(stream);
    hipEvent_t e0, e1;
    PHA_HIP(hipEventCreate(&e0));
    PHA_HIP(hipEventCreate(&e1));
    PHA_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++)
        ntt_forward(ctx->c, inout, inout, inout, plain_sel(0, cms), EPI_FWD_CANON, NttExtra{}, s);
    PHA_HIP(hipEventRecord(e1, s));
    PHA_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PHA_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / (float)iters;
    PHA_API_END
}

}  // extern "C"
