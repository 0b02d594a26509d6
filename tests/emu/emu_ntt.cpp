// emu_ntt.cpp -- TEST-ONLY host replay of the HIP NTT thread program.
//
// There is no GPU in the build container, so the exact per-thread code of
// phantom-fhe_amd/csrc/pha_ntt_core.h (index maps, LDS layout, twiddle addressing, lazy ranges,
// epilogues) is compiled for the host and executed workgroup by workgroup, segment by segment
// (a segment = the code between two barriers), thread by thread.  tests/test_emu_ntt.py compares
// the result with the oracle bit for bit.  Nothing in the product links this file.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../phantom-fhe_amd/csrc/pha_ntt_core.h"

using namespace pha;

static u64x2 g_twregs[1024][64];

template <class Prog, int SEG>
static void run_segments(const PassArgs &a, u64 *lds, u64 (*regs)[16]) {
    if constexpr (SEG == 0)
        for (int tid = 0; tid < Prog::THREADS; tid++) Prog::load_twiddles(a, tid, g_twregs[tid]);
    if constexpr (SEG < Prog::NSEG) {
        for (int tid = 0; tid < Prog::THREADS; tid++) Prog::template run<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
        run_segments<Prog, SEG + 1>(a, lds, regs);
    }
}

// the batched kernel's form of a pass (ntt_zloop_kernel): every round's twiddles requested once (HOIST 1), then per polynomial
// prefetch (the first round's global loads) + run_prefetched
static bool g_zloop_form = false;
// r05: modup_conv_s1_kernel's form of the STRIDED pass: the first round's registers are filled before the pass starts (there by the base
// conversion, here by the same global loads), every round's twiddles requested up front (HOIST 1), run_prefetched for each segment
static bool g_prefetched_strided = false;
template <class Prog, int SEG>
static void run_segments_prefetched(const PassArgs &a, u64 *lds, u64 (*regs)[16]) {
    if constexpr (SEG == 0)
        for (int tid = 0; tid < Prog::THREADS; tid++) {
            Prog::load_twiddles(a, tid, g_twregs[tid]);
            Prog::prefetch(a, tid, regs[tid]);
        }
    if constexpr (SEG < Prog::NSEG) {
        for (int tid = 0; tid < Prog::THREADS; tid++) Prog::template run_prefetched<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
        run_segments_prefetched<Prog, SEG + 1>(a, lds, regs);
    }
}

template <class C, bool FWD, int EPI, bool FOLD>
static void run_pass(PassArgs a, size_t n) {
    std::vector<u64> lds(C::LDS_WORDS);
    static u64 regs[1024][16];
    const u32 tiles = (u32)(n >> C::LOGTILE);
    for (u32 t = 0; t < tiles; t++) {
        a.tile = t;
        if ((g_zloop_form && !C::STRIDED) || (g_prefetched_strided && C::STRIDED && FWD)) run_segments_prefetched<PassProgram<C, FWD, EPI, FOLD, 1>, 0>(a, lds.data(), regs);
        else run_segments<PassProgram<C, FWD, EPI, FOLD, 2>, 0>(a, lds.data(), regs);
    }
}

template <int LOGN, int VARIANT>
static void emu(bool fwd, int epi, const u64 *in, u64 *out, u64 q, const u64x2 *tw, u64x2 ninv, u64x2 w1ninv,
                u64x2 scale, const u64 *aux, bool fp) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    const size_t n = (size_t)1 << LOGN;
    PassArgs a{};
    a.tw = tw; a.twd = reinterpret_cast<const u64 *>(tw); a.q = q; a.rho0 = P1::T; a.stride = P2::T; a.ninv = ninv; a.w1ninv = w1ninv; a.scale = scale; a.aux = aux;
    a.fp = fp; a.fpm = make_fpmod(q);
    if (fwd) {
        a.in = in; a.out = out;
        run_pass<P1, true, EPI_NONE, false>(a, n);
        a.in = out;
        if (epi == EPI_FWD_MODDOWN) run_pass<P2, true, EPI_FWD_MODDOWN, false>(a, n);
        else run_pass<P2, true, EPI_FWD_CANON, false>(a, n);
    } else {
        a.in = in; a.out = out;
        run_pass<P2, false, EPI_NONE, false>(a, n);
        a.in = out;
        if (epi == EPI_INV_SCALE) run_pass<P1, false, EPI_INV_SCALE, true>(a, n);
        else run_pass<P1, false, EPI_INV_CANON, true>(a, n);
    }
}

// Wave-locality proof for the barrier-free configurations: over all rounds, every LDS word a thread touches
// must belong to the words of its own wavefront (tid / 64) only.  Returns the number of violations.
template <class C, int RI>
static int check_round_owner(std::vector<int> &owner) {
    constexpr int r = C::r(RI), K = 1 << r, G = C::EPT >> r, s0 = C::s0(RI), LOGD = C::LOGT - s0 - r;
    int bad = 0;
    if constexpr (r > 0) {
        for (int tid = 0; tid < C::THREADS; tid++)
            for (int gi = 0; gi < G; gi++) {
                int v, hi, lo;
                decode_group<C, RI>(tid + C::THREADS * gi, v, hi, lo);
                const int e0 = (hi << (LOGD + r)) + lo;
                for (int k = 0; k < K; k++) {
                    const int idx = C::lds_index(e0 + (k << LOGD), v);
                    if (idx < 0 || idx >= C::LDS_WORDS) { bad++; continue; }
                    if (owner[idx] < 0) owner[idx] = tid >> 6;
                    else if (owner[idx] != (tid >> 6)) bad++;
                }
            }
    }
    return bad;
}
template <class C>
static int check_wave_local() {
    if (!C::WAVE_LOCAL) return 0;
    std::vector<int> owner(C::LDS_WORDS, -1);
    int bad = check_round_owner<C, 0>(owner) + check_round_owner<C, 1>(owner);
    if constexpr (C::NR >= 3) bad += check_round_owner<C, 2>(owner);
    if constexpr (C::NR >= 4) bad += check_round_owner<C, 3>(owner);
    int used = 0;
    for (int o : owner) used += o >= 0;
    if (used != C::TILE) bad += 1000000;   // every coefficient of the tile has exactly one slot
    return bad;
}
template <int LOGN, int VARIANT>
static int check_plan() {
    return check_wave_local<typename NttPlan<LOGN, VARIANT>::P1>() + check_wave_local<typename NttPlan<LOGN, VARIANT>::P2>();
}
extern "C" int emu_check_wave_local(int log_n, int variant) {
#define CHK(N) case N: return variant == 4 ? check_plan<N, 4>() : variant == 3 ? check_plan<N, 3>() : 0;
    if (variant == 5) return log_n == 14 ? check_plan<14, 5>() : log_n == 15 ? check_plan<15, 5>() : log_n == 16 ? check_plan<16, 5>() : -1;
    switch (log_n) { CHK(12) CHK(13) CHK(14) CHK(15) CHK(16) CHK(17) default: return -1; }
}
extern "C" int emu_plan_is_wave_local(int log_n, int variant) {   // bit 0: pass 1, bit 1: pass 2
#define WL(N, V) ((NttPlan<N, V>::P1::WAVE_LOCAL ? 1 : 0) | (NttPlan<N, V>::P2::WAVE_LOCAL ? 2 : 0))
#define WLC(N) case N: return variant == 4 ? WL(N, 4) : variant == 3 ? WL(N, 3) : variant == 2 ? WL(N, 2) : variant == 1 ? WL(N, 1) : WL(N, 0);
    if (variant == 5) return log_n == 14 ? WL(14, 5) : log_n == 15 ? WL(15, 5) : log_n == 16 ? WL(16, 5) : -1;
    switch (log_n) { WLC(12) WLC(13) WLC(14) WLC(15) WLC(16) WLC(17) default: return -1; }
}

// ---- the fused mod-up kernel's hand-over (modup_ip_body, pha_ntt.hip): forward contiguous pass whose last round KEEPS its outputs
// in registers (run_keep), canonical residues formed there, and the inverse transform's contiguous pass started from those
// registers (run_prefetched) -- same calls, same order, one wavefront-sized tile at a time.  ntt_out receives the canonical forward
// transform at the addresses the kernel uses for its key words (g0 + k), out the finished round trip.
template <class F, class I, int SEG>
static void keep_forward(const PassArgs &a, u64 *lds, u64 (*regs)[16]) {
    if constexpr (SEG < F::NSEG - 1) {
        for (int tid = 0; tid < F::THREADS; tid++) F::template run<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
        keep_forward<F, I, SEG + 1>(a, lds, regs);
    } else {
        for (int tid = 0; tid < F::THREADS; tid++) F::template run_keep<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
    }
}
template <class I, int SEG>
static void inverse_from_registers(const PassArgs &a, u64 *lds, u64 (*regs)[16]) {
    if constexpr (SEG < I::NSEG) {
        for (int tid = 0; tid < I::THREADS; tid++) I::template run_prefetched<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
        inverse_from_registers<I, SEG + 1>(a, lds, regs);
    }
}
template <int LOGN, int VARIANT>
static void emu_keep_inverse(const u64 *in, u64 *ntt_out, u64 *out, u64 q, const u64x2 *tw, const u64x2 *itw, u64x2 ninv, u64x2 w1ninv, bool fp) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using C = typename NttPlan<LOGN, VARIANT>::P2;
    using F = PassProgram<C, true, EPI_NONE, false, 0, false>;
    using I = PassProgram<C, false, EPI_NONE, false, 0, false>;
    constexpr int RL = C::NR - 1, r = C::r(RL), K = 1 << r, G = C::EPT >> r;
    const size_t n = (size_t)1 << LOGN;
    PassArgs a{};
    a.tw = tw; a.twd = reinterpret_cast<const u64 *>(tw); a.q = q; a.rho0 = P1::T; a.stride = C::T; a.fp = fp; a.fpm = make_fpmod(q);
    a.in = in; a.out = out;
    run_pass<P1, true, EPI_NONE, false>(a, n);       // strided pass: in -> out (the digits buffer of the kernel)
    std::vector<u64> lds(C::LDS_WORDS);
    static u64 regs[1024][16];
    const u32 tiles = (u32)(n >> C::LOGTILE);
    for (u32 t = 0; t < tiles; t++) {
        PassArgs af = a;
        af.in = out; af.tile = t;
        keep_forward<F, I, 0>(af, lds.data(), regs);
        for (int tid = 0; tid < C::THREADS; tid++)
            for (int gi = 0; gi < G; gi++) {
                int v, hi, lo;
                decode_group<C, RL>(tid + C::THREADS * gi, v, hi, lo);
                const size_t g0 = ((size_t)t * C::V + v) * C::T + ((size_t)hi << r);
                for (int kk = 0; kk < K; kk++) {
                    u64 &x = regs[tid][gi * K + kk];
                    x = fp ? fp_to_canon(as_f64(x), a.fpm) : csub(csub(csub(x, q << 2), q << 1), q);
                    ntt_out[g0 + kk] = x;
                }
            }
        PassArgs ai = a;
        ai.tw = itw; ai.twd = reinterpret_cast<const u64 *>(itw); ai.in = nullptr; ai.out = out; ai.tile = t;
        inverse_from_registers<I, 0>(ai, lds.data(), regs);
    }
    a.tw = itw; a.twd = reinterpret_cast<const u64 *>(itw); a.ninv = ninv; a.w1ninv = w1ninv;
    a.in = out; a.out = out;
    run_pass<P1, false, EPI_INV_CANON, true>(a, n);   // the strided pass alone (NttExtra::second_pass_only)
}
extern "C" int emu_keep_then_inverse(int log_n, int variant, int fp, const uint64_t *in, uint64_t *ntt_out, uint64_t *out, uint64_t q,
                                     const uint64_t *tw, const uint64_t *itw, const uint64_t *ninv, const uint64_t *w1ninv) {
    const u64x2 *t = reinterpret_cast<const u64x2 *>(tw), *it = reinterpret_cast<const u64x2 *>(itw);
    const u64x2 ni{ninv[0], ninv[1]}, w1{w1ninv[0], w1ninv[1]};
    g_zloop_form = false;
#define KI(N, V) emu_keep_inverse<N, V>(in, ntt_out, out, q, t, it, ni, w1, fp != 0)
    if (variant == 3) {
        switch (log_n) {
            case 14: KI(14, 3); return 0;
            case 15: KI(15, 3); return 0;
            case 16: KI(16, 3); return 0;
            case 17: KI(17, 3); return 0;
            default: return -1;
        }
    }
    if (variant == 5) {
        switch (log_n) {
            case 14: KI(14, 5); return 0;
            case 15: KI(15, 5); return 0;
            case 16: KI(16, 5); return 0;
            default: return -1;
        }
    }
    return -2;
}

// N = 4096 / 8192 as one pass (variant 6)
template <class P>
static void emu_whole(bool fwd, int epi, const u64 *in, u64 *out, u64 q, const u64x2 *tw, u64x2 ninv, u64x2 w1ninv,
                      u64x2 scale, const u64 *aux, bool fp) {
    PassArgs a{};
    a.tw = tw; a.twd = reinterpret_cast<const u64 *>(tw); a.q = q; a.rho0 = 1; a.stride = P::T; a.ninv = ninv; a.w1ninv = w1ninv;
    a.scale = scale; a.aux = aux; a.fp = fp; a.fpm = make_fpmod(q);
    a.in = in; a.out = out;
    if (fwd) {
        if (epi == EPI_FWD_MODDOWN) run_pass<P, true, EPI_FWD_MODDOWN, false>(a, P::T);
        else run_pass<P, true, EPI_FWD_CANON, false>(a, P::T);
    } else {
        if (epi == EPI_INV_SCALE) run_pass<P, false, EPI_INV_SCALE, true>(a, P::T);
        else run_pass<P, false, EPI_INV_CANON, true>(a, P::T);
    }
}

// bit 16 of log_n_and_variant: FP64 path (tw_interleaved is then a plain array of doubles W; ninv / w1ninv carry
// the constant as a double in their first word)
extern "C" int emu_ntt(int log_n_and_variant, int fwd, int epi, const uint64_t *in, uint64_t *out, uint64_t q,
                       const uint64_t *tw_interleaved, const uint64_t *ninv, const uint64_t *w1ninv,
                       const uint64_t *scale, const uint64_t *aux) {
    const u64x2 *tw = reinterpret_cast<const u64x2 *>(tw_interleaved);
    u64x2 ni{ninv[0], ninv[1]}, w1{w1ninv[0], w1ninv[1]}, sc{scale[0], scale[1]};
    const u64 *i = reinterpret_cast<const u64 *>(in);
    u64 *o = reinterpret_cast<u64 *>(out);
    const u64 *ax = reinterpret_cast<const u64 *>(aux);
    const int log_n = log_n_and_variant & 0xff, variant = (log_n_and_variant >> 8) & 0xff;
    const bool fp = (log_n_and_variant >> 16) & 1;
    g_zloop_form = (log_n_and_variant >> 17) & 1;   // bit 17: the contiguous pass in the batched kernel's form
    if (variant == 6) {
        if (log_n == 12) emu_whole<WholePlan12>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else if (log_n == 13) emu_whole<WholePlan13>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else if (log_n == 14) emu_whole<WholePlan14>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else return -2;
        return 0;
    }
    if (variant == 5 || variant == 7 || variant == 8) {   // four coefficients per thread (N = 2^14 .. 2^16); 7 / 8 = plans 6 / 7
#define EMU4(V) do { if (log_n == 14) emu<14, V>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else if (log_n == 15) emu<15, V>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else if (log_n == 16) emu<16, V>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else return -2; } while (0)
        if (variant == 5) EMU4(5);
        else if (variant == 7) EMU4(6);
        else EMU4(7);
        return 0;
    }
    if (variant >= 9 && variant <= 13) {   // r04 experiment plans of N = 2^16: codes 9 / 10 -> plans 8 / 9 (128 x 512), 11 / 12 -> plans 10 / 12 (64 x 1024); r05: 13 -> plan 13
        if (log_n != 16) return -2;
        if (variant == 9) emu<16, 8>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else if (variant == 10) emu<16, 9>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else if (variant == 11) emu<16, 10>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else if (variant == 12) emu<16, 12>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
        else {
            if (!fwd) return -3;   // the fused conversion only exists in front of a forward transform
            g_prefetched_strided = true;
            emu<16, 13>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp);
            g_prefetched_strided = false;
        }
        return 0;
    }
#define EMU_CASE(N) case N: if (variant == 4) emu<N, 4>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else if (variant == 3) emu<N, 3>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else if (variant == 2) emu<N, 2>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else if (variant) emu<N, 1>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); else emu<N, 0>(fwd, epi, i, o, q, tw, ni, w1, sc, ax, fp); break;
    switch (log_n) {
        EMU_CASE(12)
        EMU_CASE(13)
        EMU_CASE(14)
        EMU_CASE(15)
        EMU_CASE(16)
        EMU_CASE(17)
        default: return -1;
    }
    return 0;
}
