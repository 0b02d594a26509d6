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

static u64x2 g_twregs[512][64];

template <class Prog, int SEG>
static void run_segments(const PassArgs &a, u64 *lds, u64 (*regs)[16]) {
    if constexpr (SEG == 0)
        for (int tid = 0; tid < Prog::THREADS; tid++) Prog::load_twiddles(a, tid, g_twregs[tid]);
    if constexpr (SEG < Prog::NSEG) {
        for (int tid = 0; tid < Prog::THREADS; tid++) Prog::template run<SEG>(a, lds, tid, regs[tid], g_twregs[tid]);
        run_segments<Prog, SEG + 1>(a, lds, regs);
    }
}

template <class C, bool FWD, int EPI, bool FOLD>
static void run_pass(PassArgs a, size_t n) {
    std::vector<u64> lds(C::LDS_WORDS);
    static u64 regs[512][16];
    const u32 tiles = (u32)(n >> C::LOGTILE);
    for (u32 t = 0; t < tiles; t++) {
        a.tile = t;
        run_segments<PassProgram<C, FWD, EPI, FOLD, 2>, 0>(a, lds.data(), regs);
    }
}

template <int LOGN, int VARIANT>
static void emu(bool fwd, int epi, const u64 *in, u64 *out, u64 q, const u64x2 *tw, u64x2 ninv, u64x2 w1ninv,
                u64x2 scale, const u64 *aux, bool fp) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    const size_t n = (size_t)1 << LOGN;
    PassArgs a{};
    a.tw = tw; a.q = q; a.rho0 = P1::T; a.stride = P2::T; a.ninv = ninv; a.w1ninv = w1ninv; a.scale = scale; a.aux = aux;
    a.fp = fp; a.fpm = FpMod{(double)q, 1.0 / (double)q};
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

// bit 16 of log_n_and_variant: FP64 path (tw / ninv / w1ninv then carry (W, W/q) double bit patterns)
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
