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

#include <atomic>

namespace pha {

// tuning knob (pha_set_tuning key 0): bit 0 = 8 coefficients/thread (else 16), bit 3 = integer butterflies for every prime (FP64 path off), bit 4 = on-the-fly twiddles in the contiguous pass (implies bit 0), bit 5 = bit 4 automatically for launches of >= 1024 tiles, bit 6 = one-wavefront workgroups in the contiguous pass (NttPlan variants 3 / 4), bit 7 = N = 4096 through the two-pass plans too (default: the whole transform in one launch)
std::atomic<int> g_ntt_variant{1 | 32 | 64};  // default: 8 coefficients per thread, one-wavefront contiguous pass, on-the-fly twiddles for >= 1024 tiles
#if defined(PHA_EXP_STAMPS)
__device__ unsigned long long g_stamps[8];
__device__ unsigned long long g_wg_times[2048];
#endif
extern std::atomic<int> g_bconv_split;  // pha_rns.hip

struct NttKArgs {
    const u64 *in;
    u64 *out;
    u64 *mid;                // buffer between the two passes (pass 1: in -> mid, pass 2: mid -> out)
    const u64x2 *tw;         // table base [prime][n] (forward or inverse)
    const DModulus *mod;     // [prime]
    const u64x2 *ninv;       // [prime]
    const u64x2 *w1ninv;     // [prime]
    const u64 *twf;          // FP64 path: table base [prime][n] of doubles W (forward or inverse)
    const u64x2 *ninvf, *w1ninvf;
    const FpInfo *fpinfo;    // [prime]; null = FP64 path off
    const u64 *scale;        // [limb] or null
    const u64 *scale_shoup;  // [limb] or null
    const u64 *aux;          // fuse_moddown: cx base
    LimbSel sel;
    uint32_t log_n;
    uint32_t t1, t2;         // N = t1 * t2
    uint32_t active;         // processed limbs = sel.count minus the excluded range (pipelined kernel)
    uint32_t batch;          // polynomials per launch (blockIdx.z)
    size_t poly_stride, out_stride, aux_stride;
    uint32_t excl_step, excl_limit, excl_mod;
    const u64 *pro_src;      // rescale prologue: every limb of polynomial z reads pro_src + z * pro_stride instead
    size_t pro_stride;
};

// Per-tile arguments of limb `twr` (absolute limb index in the buffer), tile `tile`.
template <bool FWD, int EPI, bool FOLD>
__device__ __forceinline__ void tile_args(const NttKArgs &k, uint32_t twr, uint32_t tile, PassArgs &a) {
    const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
    const size_t n = (size_t)1 << k.log_n;
    a.in = k.in + (size_t)twr * n;
    a.out = k.out + (size_t)twr * n;
    a.tw = k.tw + (size_t)prime * n;
    a.twd = nullptr;
    a.q = k.mod[prime].value;
    a.tile = tile;
    a.rho0 = k.t1;
    a.stride = k.t2;
    if (!FWD && FOLD) {
        a.ninv = k.ninv[prime];
        a.w1ninv = k.w1ninv[prime];
    }
    a.pro_reduce = false;
    a.pro_ratio1 = 0;
    a.fp = false;
    if (k.fpinfo) {  // primes below 2^50 take the FP64 butterflies (uniform per workgroup)
        const FpInfo fi = k.fpinfo[prime];
        if (fi.ok) {
            a.fp = true;
            a.fpm = FpMod{fi.q, fi.qinv, (fi.ok & 2) != 0, (fi.ok & 4) != 0};
            a.twd = k.twf + (size_t)prime * n;
            if (!FWD && FOLD) {
                a.ninv = k.ninvf[prime];
                a.w1ninv = k.w1ninvf[prime];
            }
        }
    }
    if (EPI == EPI_INV_SCALE || EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD) {
        a.scale.x = k.scale[twr];
        a.scale.y = k.scale_shoup[twr];
    }
    a.aux = (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD) ? k.aux + (size_t)twr * n : nullptr;
}

#if defined(PHA_PASS_OCC)
#define PHA_PASS_ATTR __attribute__((amdgpu_waves_per_eu(PHA_PASS_OCC, PHA_PASS_OCC)))
#else
#define PHA_PASS_ATTR
#endif
template <class C, bool FWD, int EPI, bool FOLD, int HOIST>
__global__ __launch_bounds__(C::THREADS) PHA_PASS_ATTR void ntt_pass_kernel(const NttKArgs k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);

    const uint32_t twr = k.sel.start + blockIdx.y;  // limb in the buffer (uniform)
    {
        // polynomial z skips its own digit: [excl_start + z*step, min(that + len, limit))  (ntt_modup.cu:422)
        const uint32_t zd = k.excl_mod ? blockIdx.z % k.excl_mod : blockIdx.z;
        const uint32_t es = k.sel.excl_start + zd * k.excl_step;
        uint32_t ee = es + (k.sel.excl_end - k.sel.excl_start);
        ee = ee < k.excl_limit ? ee : k.excl_limit;
        if (twr >= es && twr < ee) return;
    }
    PassArgs a;
    tile_args<FWD, EPI, FOLD>(k, twr, blockIdx.x, a);
    if (k.batch > 1) {  // same limbs of several polynomials in one launch
        a.in += (size_t)blockIdx.z * k.poly_stride;
        a.out += (size_t)blockIdx.z * k.out_stride;
        if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD) a.aux += (size_t)blockIdx.z * k.aux_stride;
    }
    if (FWD && (C::STRIDED || C::WHOLE) && k.pro_src) {  // rescale prologue: transform (the last limb of polynomial z) mod this prime
        const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
        a.in = k.pro_src + (size_t)blockIdx.z * k.pro_stride;
        a.pro_reduce = true;
        a.pro_ratio1 = k.mod[prime].ratio1;
    }

    u64 reg[C::EPT];
    u64x2 twreg[C::TW_TOTAL];
    using Prog = PassProgram<C, FWD, EPI, FOLD, HOIST>;
    const int tid = threadIdx.x;
#if defined(PHA_EXP_STAMPS)   // timing experiment: cycle stamps of workgroup (0,0), wave 0
#define PHA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PHA_STAMP(i) do { } while (0)
#endif
#if defined(PHA_EXP_STAMPS)
    const unsigned long long wg_t0 = wall_clock64();
#endif
    PHA_STAMP(0);
    // The whole pass is emitted twice, once with a.fp known true and once known false (a.fp is uniform per workgroup):
    // each copy is then scheduled and register-allocated like a kernel that has only that butterfly back end
    // (sweep at 2^16, 60 / 240 / 1020 limbs: 36.3 / 118 / 505 us with one shared body, 32.8 / 107 / 458 us specialised).
    auto pass = [&](const PassArgs &pa) __attribute__((always_inline)) {
        Prog::load_twiddles(pa, tid, twreg);
        Prog::template run<0>(pa, lds, tid, reg, twreg);
        PHA_STAMP(1);
        tile_sync<C>();
        PHA_STAMP(2);
        Prog::template run<1>(pa, lds, tid, reg, twreg);
        PHA_STAMP(3);
        if constexpr (Prog::NSEG >= 3) {
            tile_sync<C>();
            PHA_STAMP(4);
            Prog::template run<2>(pa, lds, tid, reg, twreg);
            PHA_STAMP(5);
        }
        if constexpr (Prog::NSEG == 4) {
            tile_sync<C>();
            Prog::template run<3>(pa, lds, tid, reg, twreg);
        }
    };
    if (a.fp) {
        PassArgs b = a;
        b.fp = true;
        pass(b);
    } else {
        PassArgs b = a;
        b.fp = false;
        pass(b);
    }
#if defined(PHA_EXP_STAMPS)
    __builtin_amdgcn_s_waitcnt(0);
    PHA_STAMP(6);
    if (tid == 0) {  // wall-clock (100 MHz) start/end of every workgroup of the last launch
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
        if (id < 1024) { g_wg_times[2 * id] = wg_t0; g_wg_times[2 * id + 1] = wall_clock64(); }
    }
#endif
}

template <class C, bool FWD, int EPI, bool FOLD>
static void launch_pass(const NttKArgs &k, hipStream_t s) {
    const size_t n = (size_t)1 << k.log_n;
    const size_t lds_bytes = (size_t)C::LDS_WORDS * sizeof(u64);
    const unsigned tiles_per_limb = (unsigned)(n >> C::LOGTILE);
    const unsigned total = k.active * tiles_per_limb;
    // (a persistent software-pipelined form and twiddle-prefetch policies were measured and dropped:
    //  DESIGN.md section 7)
    dim3 grid(tiles_per_limb, k.sel.count, k.batch);
    (void)total;
    if (lds_bytes > 64 * 1024) {   // beyond the default dynamic LDS limit: raise it once per kernel and device
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        PHA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            PHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<C, FWD, EPI, FOLD, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((ntt_pass_kernel<C, FWD, EPI, FOLD, 0>), grid, dim3(C::THREADS), lds_bytes, s, k);
    check_launch();
}

// N = 4096 / 8192 as ONE pass (the transform fits a tile): T1 = 1, T2 = N
template <class W>
static void forward_whole(NttKArgs k, int epi, hipStream_t s) {
    k.t1 = 1;
    k.t2 = W::T;
    k.mid = k.out;
    if (epi == EPI_FWD_MODDOWN) launch_pass<W, true, EPI_FWD_MODDOWN, false>(k, s);
    else if (epi == EPI_FWD_MODDOWN_ADD) launch_pass<W, true, EPI_FWD_MODDOWN_ADD, false>(k, s);
    else launch_pass<W, true, EPI_FWD_CANON, false>(k, s);
}
template <class W>
static void inverse_whole(NttKArgs k, int epi, hipStream_t s) {
    k.t1 = 1;
    k.t2 = W::T;
    k.mid = k.out;
    if (epi == EPI_INV_SCALE) launch_pass<W, false, EPI_INV_SCALE, true>(k, s);
    else launch_pass<W, false, EPI_INV_CANON, true>(k, s);
}

template <int LOGN, int VARIANT>
static void forward_impl(NttKArgs k, int epi, hipStream_t s) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    u64 *const final_out = k.out;
    const size_t final_stride = k.out_stride;
    k.out = k.mid;
    k.out_stride = k.poly_stride;
    launch_pass<P1, true, EPI_NONE, false>(k, s);
    k.in = k.mid;
    k.out = final_out;
    k.out_stride = final_stride;
    // pass 1 ran in -> mid with the input stride; pass 2 reads mid and writes out with the output stride
    if (epi == EPI_FWD_MODDOWN) launch_pass<P2, true, EPI_FWD_MODDOWN, false>(k, s);
    else if (epi == EPI_FWD_MODDOWN_ADD) launch_pass<P2, true, EPI_FWD_MODDOWN_ADD, false>(k, s);
    else launch_pass<P2, true, EPI_FWD_CANON, false>(k, s);
}

template <int LOGN, int VARIANT>
static void inverse_impl(NttKArgs k, int epi, hipStream_t s) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    u64 *const final_out = k.out;
    const size_t final_stride = k.out_stride;
    k.out = k.mid;
    k.out_stride = k.poly_stride;
    launch_pass<P2, false, EPI_NONE, false>(k, s);
    k.in = k.mid;
    k.out = final_out;
    k.out_stride = final_stride;
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
    const bool use_fp = !(g_ntt_variant.load(std::memory_order_relaxed) & 8);
    k.twf = fwd ? c.d_twf.p : c.d_itwf.p;
    k.ninvf = c.d_ninvf.p;
    k.w1ninvf = c.d_w1ninvf.p;
    k.fpinfo = use_fp ? c.d_fpinfo.p : nullptr;
    k.scale = x.scale;
    k.scale_shoup = x.scale_shoup;
    k.aux = x.aux;
    k.sel = sel;
    k.log_n = c.log_n;
    k.batch = x.batch ? x.batch : 1;
    k.poly_stride = x.poly_stride;
    k.out_stride = x.out_stride ? x.out_stride : x.poly_stride;
    k.aux_stride = x.aux_stride ? x.aux_stride : x.poly_stride;
    k.excl_step = x.excl_step;
    k.excl_limit = x.excl_limit;
    k.excl_mod = x.excl_mod;
    k.pro_src = fwd ? x.pro_src : nullptr;
    k.pro_stride = x.pro_stride;
    uint32_t excl = 0;
    if (sel.excl_end > sel.excl_start) {
        const uint32_t lo = sel.excl_start > sel.start ? sel.excl_start : sel.start;
        const uint32_t hi = sel.excl_end < sel.start + sel.count ? sel.excl_end : sel.start + sel.count;
        excl = hi > lo ? hi - lo : 0;
        if (excl && (lo != sel.excl_start || hi != sel.excl_end)) excl = 0xffffffffu;  // partial overlap: not pipelined
    }
    k.active = excl == 0xffffffffu ? 0 : sel.count - excl;
    return k;
}

static void check_sel(Context &c, const LimbSel &sel) {
    if (sel.count == 0) return;
    const uint32_t last = sel.start + sel.count - 1;
    const uint32_t prime_last = last >= sel.remap_from ? last + sel.remap_add : last;
    if (prime_last >= c.rows) throw std::invalid_argument("modulus index out of range of the NTT tables");
}

void ntt_forward(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, true);
    const int vv = g_ntt_variant.load(std::memory_order_relaxed);
    // bit 5 (default): large launches (>= 1024 tiles: the memory-bound throughput regime) take the on-the-fly
    // twiddle plan; small ones are latency-bound and keep the table-driven last round (r01c measurements)
    const size_t tiles = ((size_t)c.n / kTileElems) * sel.count * (x.batch ? x.batch : 1);
    const bool ot = (vv & 16) || ((vv & 32) && (vv & 1) && tiles >= 1024), wave = (vv & 64) && (vv & 1);
    const int v = ot ? (wave ? 4 : 2) : wave ? 3 : (vv & 1);
    if (c.log_n == 12 && !(vv & 128)) {  // bit 7 clear (default): N = 4096 in one launch
        forward_whole<WholePlan12>(k, epi, s);
        return;
    }
    // N = 8192 in one launch pays from about 64 limb-polynomials per launch (one 512-thread workgroup per limb: 1 / 10 /
    // 60 / 240 / 1020 limbs 10.4 / 10.8 / 11.5 / 13.7 / 53.8 us against 8.7 / 9.4 / 12.0 / 26.2 / 58.2 in two passes);
    // bit 8 forces it for every size (tests)
    if (c.log_n == 13 && !(vv & 128) && ((vv & 256) || (size_t)sel.count * (x.batch ? x.batch : 1) >= 64)) {
        forward_whole<WholePlan13>(k, epi, s);
        return;
    }
    switch (c.log_n) {
        case 12: if (v == 4) forward_impl<12, 4>(k, epi, s); else if (v == 3) forward_impl<12, 3>(k, epi, s); else if (v == 2) forward_impl<12, 2>(k, epi, s); else if (v) forward_impl<12, 1>(k, epi, s); else forward_impl<12, 0>(k, epi, s); break;
        case 13: if (v == 4) forward_impl<13, 4>(k, epi, s); else if (v == 3) forward_impl<13, 3>(k, epi, s); else if (v == 2) forward_impl<13, 2>(k, epi, s); else if (v) forward_impl<13, 1>(k, epi, s); else forward_impl<13, 0>(k, epi, s); break;
        case 14: if (v == 4) forward_impl<14, 4>(k, epi, s); else if (v == 3) forward_impl<14, 3>(k, epi, s); else if (v == 2) forward_impl<14, 2>(k, epi, s); else if (v) forward_impl<14, 1>(k, epi, s); else forward_impl<14, 0>(k, epi, s); break;
        case 15: if (v == 4) forward_impl<15, 4>(k, epi, s); else if (v == 3) forward_impl<15, 3>(k, epi, s); else if (v == 2) forward_impl<15, 2>(k, epi, s); else if (v) forward_impl<15, 1>(k, epi, s); else forward_impl<15, 0>(k, epi, s); break;
        case 16: if (v == 4) forward_impl<16, 4>(k, epi, s); else if (v == 3) forward_impl<16, 3>(k, epi, s); else if (v == 2) forward_impl<16, 2>(k, epi, s); else if (v) forward_impl<16, 1>(k, epi, s); else forward_impl<16, 0>(k, epi, s); break;
        case 17: if (v == 4) forward_impl<17, 4>(k, epi, s); else if (v == 3) forward_impl<17, 3>(k, epi, s); else if (v == 2) forward_impl<17, 2>(k, epi, s); else if (v) forward_impl<17, 1>(k, epi, s); else forward_impl<17, 0>(k, epi, s); break;
        default: throw std::invalid_argument("unsupported polynomial degree");
    }
}

void ntt_inverse(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, false);
    const int vv = g_ntt_variant.load(std::memory_order_relaxed);
    // bit 5 (default): large launches (>= 1024 tiles: the memory-bound throughput regime) take the on-the-fly
    // twiddle plan; small ones are latency-bound and keep the table-driven last round (r01c measurements)
    const size_t tiles = ((size_t)c.n / kTileElems) * sel.count * (x.batch ? x.batch : 1);
    const bool ot = (vv & 16) || ((vv & 32) && (vv & 1) && tiles >= 1024), wave = (vv & 64) && (vv & 1);
    const int v = ot ? (wave ? 4 : 2) : wave ? 3 : (vv & 1);
    if (c.log_n == 12 && !(vv & 128)) {
        inverse_whole<WholePlan12>(k, epi, s);
        return;
    }
    if (c.log_n == 13 && !(vv & 128) && ((vv & 256) || (size_t)sel.count * (x.batch ? x.batch : 1) >= 64)) {
        inverse_whole<WholePlan13>(k, epi, s);
        return;
    }
    switch (c.log_n) {
        case 12: if (v == 4) inverse_impl<12, 4>(k, epi, s); else if (v == 3) inverse_impl<12, 3>(k, epi, s); else if (v == 2) inverse_impl<12, 2>(k, epi, s); else if (v) inverse_impl<12, 1>(k, epi, s); else inverse_impl<12, 0>(k, epi, s); break;
        case 13: if (v == 4) inverse_impl<13, 4>(k, epi, s); else if (v == 3) inverse_impl<13, 3>(k, epi, s); else if (v == 2) inverse_impl<13, 2>(k, epi, s); else if (v) inverse_impl<13, 1>(k, epi, s); else inverse_impl<13, 0>(k, epi, s); break;
        case 14: if (v == 4) inverse_impl<14, 4>(k, epi, s); else if (v == 3) inverse_impl<14, 3>(k, epi, s); else if (v == 2) inverse_impl<14, 2>(k, epi, s); else if (v) inverse_impl<14, 1>(k, epi, s); else inverse_impl<14, 0>(k, epi, s); break;
        case 15: if (v == 4) inverse_impl<15, 4>(k, epi, s); else if (v == 3) inverse_impl<15, 3>(k, epi, s); else if (v == 2) inverse_impl<15, 2>(k, epi, s); else if (v) inverse_impl<15, 1>(k, epi, s); else inverse_impl<15, 0>(k, epi, s); break;
        case 16: if (v == 4) inverse_impl<16, 4>(k, epi, s); else if (v == 3) inverse_impl<16, 3>(k, epi, s); else if (v == 2) inverse_impl<16, 2>(k, epi, s); else if (v) inverse_impl<16, 1>(k, epi, s); else inverse_impl<16, 0>(k, epi, s); break;
        case 17: if (v == 4) inverse_impl<17, 4>(k, epi, s); else if (v == 3) inverse_impl<17, 3>(k, epi, s); else if (v == 2) inverse_impl<17, 2>(k, epi, s); else if (v) inverse_impl<17, 1>(k, epi, s); else inverse_impl<17, 0>(k, epi, s); break;
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
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                          size_t start, size_t size_QP, size_t size_P,
                                                          void *stream) {
    PHA_CTX_BEGIN(ctx)
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
    PHA_CTX_BEGIN(ctx)
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
    PHA_CTX_BEGIN(ctx)
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
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_inverse(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms, size_t start,
                               void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(out); need(in);
    ntt_inverse(ctx->c, in, out, out, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_scale(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms,
                                     size_t start, const uint64_t *scale, const uint64_t *scale_shoup,
                                     void *stream) {
    PHA_CTX_BEGIN(ctx)
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
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    ntt_inverse(ctx->c, inout, inout, inout, special_sel(start, cms, size_QP, size_P), EPI_INV_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

// BEHZ base Bsk = B u {m_sk} (src/evaluate.cu:434,528).  The reference keeps the tables of Bsk u {m_tilde} in their own
// DNTTTable and sends the last data limb (m_sk) to its last row (fntt_2d.cu:226, intt_2d.cu:334); here those primes
// are auxiliary rows of the context's one table set, so the selector is a plain offset.
static LimbSel temp_mod_sel(Context &c, size_t cms, size_t start, size_t total) {
    Behz &b = c.behz();
    if (start != 0 || cms != b.size_bsk || total != (size_t)b.size_bsk + 1)
        throw std::invalid_argument("include_temp_mod transforms a whole Bsk buffer: coeff_modulus_size = |Bsk|, total = |Bsk| + 1");
    LimbSel s = plain_sel(0, cms);
    s.remap_from = 0;
    s.remap_add = b.aux0;
    return s;
}

int pha_nwt_2d_radix8_forward_inplace_include_temp_mod(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                                       size_t total_modulus_size, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_forward(ctx->c, inout, inout, inout, temp_mod_sel(ctx->c, cms, start, total_modulus_size), EPI_FWD_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                              size_t start, size_t total_modulus_size,
                                                              const uint64_t *scale, const uint64_t *scale_shoup,
                                                              void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout); need(scale); need(scale_shoup);
    NttExtra x;
    x.scale = scale;
    x.scale_shoup = scale_shoup;
    ntt_inverse(ctx->c, inout, inout, inout, temp_mod_sel(ctx->c, cms, start, total_modulus_size), EPI_INV_SCALE, x,
                as_stream(stream));
    PHA_API_END
}

// out[limb] = NTT modulo q_{modulus_index} of in[limb] for the limbs [start, start + cms): the reference lifts a
// plaintext (coefficients below t < q) into one RNS limb at a time this way (ntt_keyswitch_old.cu:225-265, callers
// evaluate.cu:1152,1210,1321).  Inputs are reduced modulo the prime as they are loaded (the reference leaves
// that reduction commented out, :47-49: same results for its inputs below q).
int pha_nwt_2d_radix8_forward_modup_fuse(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t modulus_index,
                                         size_t cms, size_t start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(out); need(in);
    Context &c = ctx->c;
    if (modulus_index >= c.size_qp) throw std::invalid_argument("modulus_index out of range");
    for (size_t limb = start; limb < start + cms; limb++) {
        LimbSel sel = plain_sel(limb, 1);
        sel.remap_from = (uint32_t)limb;
        sel.remap_add = (uint32_t)modulus_index - (uint32_t)limb;
        NttExtra x;
        x.pro_src = in + limb * c.n;
        ntt_forward(c, out, out, out, sel, EPI_FWD_CANON, x, as_stream(stream));
    }
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                              size_t batch, size_t poly_stride, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (batch == 0 || batch > 65535) throw std::invalid_argument("batch out of range");
    NttExtra x;
    x.batch = (uint32_t)batch;
    x.poly_stride = poly_stride;
    ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                               size_t batch, size_t poly_stride, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (batch == 0 || batch > 65535) throw std::invalid_argument("batch out of range");
    NttExtra x;
    x.batch = (uint32_t)batch;
    x.poly_stride = poly_stride;
    ntt_inverse(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_INV_CANON, x, as_stream(stream));
    PHA_API_END
}

#if defined(PHA_EXP_STAMPS)
int pha_exp_read_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
}
int pha_exp_read_wg_times(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 2048);
}
#endif

int pha_set_tuning(int key, int value) {
    PHA_API_BEGIN
    if (key == 0) {
        if (value < 0 || value > 511 || (value & 6)) throw std::invalid_argument("unknown NTT variant");
        g_ntt_variant.store(value);
    } else if (key == 1) {
        g_bconv_split.store(value ? 1 : 0);
    } else {
        throw std::invalid_argument("unknown tuning key");
    }
    PHA_API_END
}

int pha_time_forward_ntt(pha_context_t ctx, uint64_t *inout, size_t cms, int iters, void *stream, float *ms_out) {
    PHA_CTX_BEGIN(ctx)
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
