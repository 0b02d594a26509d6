// pha_ntt_core.h -- workgroup-tile NTT passes for gfx950.
//
// What it computes (reference semantics): the in-place negacyclic NTT of
// src/ntt/fntt_2d.cu:9-198,620-653 (forward: natural in -> bit-reversed out, canonical [0,q)) and
// src/ntt/intt_2d.cu:9-311,724-794 (inverse: bit-reversed in -> natural out, times N^-1), i.e. the
// SEAL-order Cooley-Tukey / Gentleman-Sande stage loops with twiddle tw[m + i] for group i of the
// stage with m groups (src/ntt/ntt_1d.cu:50, include/butterfly.cuh:39-96).
//
// How (MI355X-first, not the reference's 2-phase radix-8 split):
//   * N = T1 * T2.  Two passes, each one workgroup (256 threads = 4 wavefronts) per 4096-element
//     tile held in 32 KiB of LDS, 16 coefficients per thread in registers:
//       strided pass  : T1-point transforms along stride T2 (first log2 T1 forward stages); a tile
//                       is T1 rows x V=4096/T1 adjacent columns, so every global access is a run
//                       of V*8 >= 128 contiguous bytes and every twiddle is shared by all columns.
//       contiguous pass: T2-point transforms on V=4096/T2 whole rows (a contiguous 32 KiB chunk).
//   * Inside a pass the log2 T stages are grouped into 2-3 register rounds of radix 16/8
//     (4/3 stages each); rounds exchange data through padded LDS, one barrier per exchange.
//   * The limb (RNS prime) index is uniform per workgroup (blockIdx.y), so modulus, 2q and the
//     table base live in SGPRs and no per-thread divide/modulo by N is needed.
//   * Twiddles are stored interleaved (w, w') so one 16-byte load feeds one butterfly.
//
// Everything here is plain templated C++ that also compiles for the host so tests/emu can replay
// the exact thread program (indexing, LDS layout, twiddle addressing) on the CPU against the oracle.
#pragma once
#include "pha_arith.h"

namespace pha {

constexpr int kTileElems = 4096;  // coefficients per workgroup tile = 32 KiB of LDS

// Epilogues of the last round of a transform (what is stored to global memory).
enum Epilogue {
    EPI_NONE = 0,         // intermediate pass: store lazy values
    EPI_FWD_CANON = 1,    // forward: csub 2q, csub q            (fntt_2d.cu:187-193)
    EPI_FWD_MODDOWN = 2,  // forward: out = (cx - NTT(delta)) * PInv  (ntt_moddown.cu:203-208)
    EPI_INV_CANON = 3,    // inverse: csub q                     (intt_2d.cu:201-205)
    EPI_INV_SCALE = 4,    // inverse: full Shoup multiply by per-limb scale (intt_2d.cu:305-309)
    EPI_FWD_MODDOWN_ADD = 5,  // forward: out += (cx - NTT(delta)) * PInv  (mod-down fused with add_to_ct_kernel)
    // forward: out = (ct + cx * PInv - NTT(v)) * q_last^-1 -- the mod-down epilogue (ntt_moddown.cu:203-208), add_to_ct
    // (rns_bconv.cu:763-769) and the rescale epilogue (rns.cu:1141-1158) of key switch + rescale in one (pha_keyswitch_rescale)
    EPI_FWD_KSRESCALE = 6,
};

// Round schedule of one pass: LOGT stages split into NR rounds of R0,R1,R2 stages (forward order).
// EPT_ = coefficients per thread (16: 256-thread workgroups, radix-16 rounds; 8: 512-thread
// workgroups, radix-8 rounds, twice the wavefronts per tile).
// OT_ (contiguous pass only): the twiddles of the pass's LAST round (the 2-3 finest stages, which hold
// 75-88 % of the table bytes) are not streamed from their own table rows but formed on the fly from
//     tw[(T1 + r) * 2^s + i] = tw[(T1 + r) * 2^s] * tw[i]   (mod q)
// i.e. one per-row factor and one entry of the first T2/2 table slots shared by every row; the
// butterfly multiplies by the two factors in turn.  Costs one extra modular multiply per butterfly in
// that round, removes most of the twiddle traffic of the memory-bound contiguous pass.
// LOGTILE_: log2 of the coefficients one workgroup owns.  12 = the 4096-coefficient tile (32 KiB of LDS).
// A contiguous pass may use 9: 512 coefficients = 64 threads x 8, i.e. one wavefront per workgroup that owns
// whole rows, so the exchanges between rounds stay inside the wavefront and its barriers compile away.
// (Splitting the STRIDED pass by columns between the wavefronts makes it barrier-free too; measured r01g:
// 16-byte runs per row cost more than the barriers, 45 limbs 29.1 -> 34.1 us, so that form was dropped.)
// WHOLE_: the pass is the whole transform (N = 4096 fits one tile): it is both the first and the last pass.
// R3_: a fourth round (whole-transform plans beyond 12 stages).
template <int LOGT_, bool STRIDED_, int R0_, int R1_, int R2_ = 0, int EPT_ = 16, bool OT_ = false, int LOGTILE_ = 12,
          bool WHOLE_ = false, int R3_ = 0>
struct PassCfg {
    static constexpr int EPT = EPT_;
    static constexpr bool OT = OT_ && !STRIDED_;
    static constexpr int LOGTILE = LOGTILE_;
    static constexpr int TILE = 1 << LOGTILE_;
    static constexpr int THREADS = TILE / EPT_;
    static_assert(LOGTILE_ >= LOGT_ && LOGTILE_ <= 14, "a tile holds whole transforms and fits the LDS budget (2^14 words + padding = 144 KiB of the 160 KiB)");
    // one wavefront per workgroup: every exchange between rounds stays inside it, no workgroup barrier
    static constexpr bool WAVE_LOCAL = TILE / EPT_ == 64;
    static constexpr bool WHOLE = WHOLE_ && !STRIDED_;
    static_assert((1 << R0_) <= EPT_ && (1 << R1_) <= EPT_ && (1 << R2_) <= EPT_ && (1 << R3_) <= EPT_, "radix exceeds registers");
    static_assert(!R3_ || R2_, "rounds are filled in order");
    static constexpr int LOGT = LOGT_;
    static constexpr int T = 1 << LOGT_;
    static constexpr int LOGV = LOGTILE_ - LOGT_;
    static constexpr int V = 1 << LOGV;
    static constexpr bool STRIDED = STRIDED_;
    static constexpr int NR = R3_ ? 4 : R2_ ? 3 : 2;
    static_assert(R0_ + R1_ + R2_ + R3_ == LOGT_, "round schedule must cover all stages");
    static_assert(LOGT_ >= 4 && LOGT_ <= 14, "tile transform length out of range");
    static_assert(THREADS >= 64, "a workgroup is at least one wavefront");
    static constexpr int r(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : i == 2 ? R2_ : R3_; }
    // twiddle registers per thread: round i holds G_i groups x (2^r_i - 1) pairs
    static constexpr bool ot_round(int i) { return OT && i == NR - 1; }
    // on-the-fly round: (2^r - 1) shared factors + r per-row factors for each of the G groups
    static constexpr int tw_count(int i) {
        return !r(i) ? 0 : ot_round(i) ? ((1 << r(i)) - 1) + (EPT_ >> r(i)) * r(i) : (EPT_ >> r(i)) * ((1 << r(i)) - 1);
    }
    static constexpr int tw_off(int i) {
        return i == 0 ? 0 : i == 1 ? tw_count(0) : i == 2 ? tw_count(0) + tw_count(1) : tw_count(0) + tw_count(1) + tw_count(2);
    }
    static constexpr int TW_TOTAL = tw_count(0) + tw_count(1) + tw_count(2) + tw_count(3);
    static constexpr int s0(int i) { return i == 0 ? 0 : i == 1 ? R0_ : i == 2 ? R0_ + R1_ : R0_ + R1_ + R2_; }
    // LDS layouts (u64 units).  contiguous pass: [v][e] with 2 words of padding per 16 so that a
    // thread's 16-coefficient run starts on a distinct 16-byte bank slot (ds_read_b128, 16-lane
    // groups).  strided pass: [e][v] with one 128-byte skew per 16 rows (ds_read_b64).
    static constexpr int PITCH = T + ((T >> 4) << 1);
    static constexpr int LDS_WORDS = STRIDED_ ? (T * V + (T >> 4) * 16) : (V * PITCH);
    static constexpr int lds_index(int e, int v) {
        return STRIDED_ ? (e * V + v + ((e >> 4) << 4)) : (v * PITCH + e + ((e >> 4) << 1));
    }
};

#ifndef PHA_FPSEL
#define PHA_FPSEL(a) ((a).fp)   // experiment hook: -D'PHA_FPSEL(a)=true' builds FP64-only kernels (timing only)
#endif
// Per-workgroup arguments (all uniform across the workgroup -> SGPRs).
struct PassArgs {
    const u64 *in;     // limb base to read (first round only)
    u64 *out;          // limb base to write (last round only)
    const u64x2 *tw;   // this prime's twiddle row: forward table or inverse table (integer path: W and Shoup quotient)
    const u64 *twd;    // FP64 path: the same row as doubles (W only, 8 bytes per entry)
    u64 q;             // modulus
    u32 tile;          // tile index inside the limb
    u32 rho0;          // contiguous pass: T1 (root index of row r is T1 + r); strided pass: unused
    u32 stride;        // strided pass: distance in elements between consecutive rows (= T2)
    // inverse last stage: N^-1 and itw[1]*N^-1 (intt_2d.cu:195-198 / host/ntt.cu:53-55)
    u64x2 ninv, w1ninv;
    // epilogue operands
    u64x2 scale;       // EPI_INV_SCALE: per-limb scale ; EPI_FWD_MODDOWN: PInv mod q ; EPI_FWD_KSRESCALE: q_last^-1 mod q
    const u64 *aux;    // EPI_FWD_MODDOWN: cx limb base
    u64x2 scale2;      // EPI_FWD_KSRESCALE: PInv mod q
    const u64 *aux2;   // EPI_FWD_KSRESCALE: ct limb base
    // FP64 path (q < 2^50): twd / ninv.x / w1ninv.x then hold W as a double; registers, LDS and the
    // inter-pass buffer carry doubles; global inputs and outputs stay canonical integers
    bool fp;
    FpMod fpm;
    // rescale prologue (first pass of a forward transform): `in` is ANOTHER limb's coefficients, which are
    // first reduced modulo this limb's prime (divide_and_round_reduce_q_last_kernel rns.cu:1128-1139)
    bool pro_reduce;
    u64 pro_ratio1;    // floor(2^64 / q)
};

template <class C>
PHA_HD size_t global_index(const PassArgs &a, int e, int v) {
    if (C::STRIDED) return (size_t)e * a.stride + (size_t)a.tile * C::V + v;
    return ((size_t)a.tile * C::V + v) * C::T + e;
}

// Decode radix-group g of round RI into (vector lane v, high part hi, low part lo):
// element e(k) = hi * (D << r) + k * D + lo with D = T >> (s0 + r).
template <class C, int RI>
PHA_HD void decode_group(int g, int &v, int &hi, int &lo) {
    constexpr int r = C::r(RI), s0 = C::s0(RI), LOGD = C::LOGT - s0 - r;
    if (C::STRIDED) {
        v = g & (C::V - 1);
        int rest = g >> C::LOGV;
        lo = rest & ((1 << LOGD) - 1);
        hi = rest >> LOGD;
    } else {
        lo = g & ((1 << LOGD) - 1);
        int rest = g >> LOGD;
        hi = rest & ((1 << s0) - 1);
        v = rest >> s0;
    }
}

// Twiddles of one radix-2^R group, preloaded in heap order: stage j, sub-group kk -> t[(1 << j) - 1 + kk],
// taken from tw[(base0 << j) + kk] (base0 = rho * 2^s0 + hi, see round_compute).
// One table entry: the integer path reads the (W, W') pair, the FP64 path the double W (PHA_FPSEL(a) is uniform).
PHA_HD u64x2 tw_entry(const PassArgs &a, u32 idx) {
    if (PHA_FPSEL(a)) return u64x2{a.twd[idx], 0};
    return a.tw[idx];
}
template <int R>
PHA_HD void load_group_twiddles(u64x2 *t, const PassArgs &a, u32 base0) {
    if (PHA_FPSEL(a)) {
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int kk = 0; kk < (1 << j); kk++) t[(1 << j) - 1 + kk].x = a.twd[(base0 << j) + kk];
        return;
    }
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
        for (int kk = 0; kk < (1 << j); kk++) t[(1 << j) - 1 + kk] = a.tw[(base0 << j) + kk];
}

// r forward stages on 2^r registers; stage j uses tw[(base0 << j) + (k >> (r - j))].
template <int R>
PHA_HD void ct_round(u64 *v, const u64x2 *t, u64 q4, u64 nq) {
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int dist = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            const int kk = k >> (R - j);
            ct_bfly4(v[k], v[k + dist], t[(1 << j) - 1 + kk], q4, nq);
        }
    }
}

// On-the-fly forms: tc = shared factors in heap order (stage j, sub-group kk -> tc[(1<<j)-1+kk]),
// tr[j] = the row factor of stage j.  Y is multiplied by tr[j], then by tc[...].
template <int R, bool FWD>
PHA_HD void ot_round_int(u64 *v, const u64x2 *tc, const u64x2 *tr, u64 q4, u64 nq) {
#pragma unroll
    for (int jj = 0; jj < R; jj++) {
        const int j = FWD ? jj : R - 1 - jj;
        const int dist = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            const u64x2 wc = tc[(1 << j) - 1 + (k >> (R - j))], wr = tr[j];
            u64 &X = v[k], &Y = v[k + dist];
            if (FWD) {
                const u64 x = csub(X, q4);
                const u64 t = shoup_lazy4(shoup_lazy4(Y, wr, nq), wc, nq);
                X = x + t;
                Y = x + q4 - t;
            } else {
                const u64 s = X + Y, d = X + q4 - Y;
                X = csub(s, q4);
                Y = shoup_lazy4(shoup_lazy4(d, wr, nq), wc, nq);
            }
        }
    }
}
template <int R, bool FWD, bool LIGHT>
PHA_HD void ot_round_fp(u64 *v, const u64x2 *tc, const u64x2 *tr, FpMod m) {
#pragma unroll
    for (int jj = 0; jj < R; jj++) {
        const int j = FWD ? jj : R - 1 - jj;
        const int dist = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            const u64x2 wc = tc[(1 << j) - 1 + (k >> (R - j))], wr = tr[j];
            double X = as_f64(v[k]), Y = as_f64(v[k + dist]);
            if (FWD) {
                const double t = LIGHT ? fp_mulmod_light(fp_mulmod_light(Y, as_f64(wr.x), m), as_f64(wc.x), m)
                                       : fp_mulmod(fp_mulmod(Y, as_f64(wr.x), m), as_f64(wc.x), m);
                const double x = X;
                X = x + t;
                Y = x - t;
            } else {
                const double s = X + Y, d = X - Y;
                X = LIGHT ? s : fp_reduce(s, m);
                Y = LIGHT ? fp_mulmod_light(fp_mulmod_light(d, as_f64(wr.x), m), as_f64(wc.x), m)
                          : fp_mulmod(fp_mulmod(d, as_f64(wr.x), m), as_f64(wc.x), m);
            }
            v[k] = as_u64(X);
            v[k + dist] = as_u64(Y);
        }
    }
}

// FP64 form of the forward round (same twiddle order; t[i].x = W as a double): LIGHT butterflies for every prime below 2^50 -- the
// registers are re-centred by the caller after the rounds PassProgram::fp_sched() names (r04; r01-r03 re-centred every product of
// primes from 2^47 inside the butterfly, fp_ct_bfly)
template <int R>
PHA_HD void fp_ct_round(u64 *v, const u64x2 *t, FpMod m) {
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int dist = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            const u64x2 w = t[(1 << j) - 1 + (k >> (R - j))];
            const double X = as_f64(v[k]), tt = fp_mulmod_light(as_f64(v[k + dist]), as_f64(w.x), m);
            v[k] = as_u64(X + tt);
            v[k + dist] = as_u64(X - tt);
        }
    }
}
// Which inverse stages must re-centre the sum X + Y (primes up to 2^50, magnitudes must stay below 2^52.6 ~ 6q):
// with inputs bounded by b*q a stage forms |X - Y| <= 2b q for the multiply (needs 2b <= 6) and leaves X + Y
// <= 2b q behind, so the sum may stay unreduced while 2b <= 3.  From centred inputs (b = 1/2: every pass but the
// first, and anything a multiply produced) that is two stages in three; canonical inputs in [0, q) (b = 1, first
// pass) allow one first.  `i` counts the stages of the pass in execution order.
constexpr bool gs_stage_reduces(int i, bool canon_in) { return canon_in ? (i % 3 == 1) : (i % 3 == 2); }

// STAGE0: how many stages of this pass ran before this round; CANON_IN: the pass started from canonical integers
template <int R, bool FOLD, bool LIGHT, int STAGE0 = 0, bool CANON_IN = true>
PHA_HD void fp_gs_round(u64 *v, const u64x2 *t, FpMod m, u64x2 ninv, u64x2 w1ninv) {
#pragma unroll
    for (int j = R - 1; j >= 0; j--) {
        const int dist = 1 << (R - 1 - j);
        const bool reduce_sum = gs_stage_reduces(STAGE0 + (R - 1 - j), CANON_IN);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            double X = as_f64(v[k]), Y = as_f64(v[k + dist]);
            if (FOLD && j == 0) {
                const double s = X + Y, d = X - Y;
                X = fp_mulmod(s, as_f64(ninv.x), m);
                Y = fp_mulmod(d, as_f64(w1ninv.x), m);
            } else {
                const u64x2 w = t[(1 << j) - 1 + (k >> (R - j))];
                if (LIGHT) {
                    const double s = X + Y, d = X - Y;
                    X = s;
                    Y = fp_mulmod_light(d, as_f64(w.x), m);
                } else if (reduce_sum) {
                    fp_gs_bfly(X, Y, as_f64(w.x), m);
                } else {
                    const double s = X + Y, d = X - Y;
                    X = s;
                    Y = fp_mulmod(d, as_f64(w.x), m);
                }
            }
            v[k] = as_u64(X);
            v[k + dist] = as_u64(Y);
        }
    }
}

// r inverse stages (reverse order).  FOLD: the j == 0 stage is the transform's last stage (m = 1)
// and carries N^-1: X' = (X+Y)*ninv, Y' = (X-Y)*(itw[1]*ninv), both lazy [0,4q).
template <int R, bool FOLD>
PHA_HD void gs_round(u64 *v, const u64x2 *t, u64 q4, u64 nq, u64x2 ninv, u64x2 w1ninv) {
#pragma unroll
    for (int j = R - 1; j >= 0; j--) {
        const int dist = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & dist) continue;
            const int kk = k >> (R - j);
            if (FOLD && j == 0) {
                u64 s = v[k] + v[k + dist];
                u64 d = v[k] + q4 - v[k + dist];
                v[k] = shoup_lazy4(s, ninv, nq);
                v[k + dist] = shoup_lazy4(d, w1ninv, nq);
            } else {
                gs_bfly4(v[k], v[k + dist], t[(1 << j) - 1 + kk], q4, nq);
            }
        }
    }
}

// value-only form: aux / accumulate operands are passed in (so the caller can fetch them with 16-byte loads)
// Epilogues that the FP64 path computes on the doubles themselves (r04): the integer forms cost a Shoup multiply (~17 vector
// instructions on 64-bit halves) per product on top of the conversion; here x arrives as the last round's lazy double (below 7.5 q),
// the operands are canonical words below 2^50, every product is an exact fp_mulmod_light (inputs below 2.4 q), and one fp_to_canon
// ends it -- the same residues, bit for bit.
constexpr bool fp_epilogue(int epi) {
    return epi == EPI_FWD_MODDOWN || epi == EPI_FWD_MODDOWN_ADD || epi == EPI_FWD_KSRESCALE || epi == EPI_INV_SCALE;
}
template <int EPI>
PHA_HD u64 apply_epilogue_fp(u64 x, const PassArgs &a, u64 aux, u64 acc) {
    const FpMod m = a.fpm;
    if (EPI == EPI_INV_SCALE) return fp_to_canon(fp_mulmod_light(as_f64(x), fp_from_canon(a.scale.x), m), m);   // |x| < 8 q -> |.| < 3.5 q
    const double t = fp_reduce(as_f64(x), m);                                   // NTT value, centred
    if (EPI == EPI_FWD_KSRESCALE) {   // (ct + cx * PInv - t) * q_last^-1 ; acc carries the ct word
        const double u = fp_from_canon(acc) + fp_mulmod_light(fp_from_canon(aux), fp_from_canon(a.scale2.x), m);   // < 1.9 q
        return fp_to_canon(fp_mulmod_light(u - t, fp_from_canon(a.scale.x), m), m);
    }
    const double r = fp_mulmod_light(fp_from_canon(aux) - t, fp_from_canon(a.scale.x), m);   // (cx - t) * PInv, input < 1.6 q
    return fp_to_canon(EPI == EPI_FWD_MODDOWN_ADD ? r + fp_from_canon(acc) : r, m);
}
template <int EPI>
PHA_HD u64 apply_epilogue_v(u64 x, const PassArgs &a, u64 aux, u64 acc) {
    const u64 q = a.q;
    if (fp_epilogue(EPI) && PHA_FPSEL(a)) return apply_epilogue_fp<EPI>(x, a, aux, acc);   // (uniform)
    // otherwise the FP64 path hands over canonical residues (fp_to_canon in fp_before_global_store); PHA_FPSEL(a) is uniform per workgroup
    if (EPI == EPI_FWD_CANON) return PHA_FPSEL(a) ? x : csub(csub(csub(x, q << 2), q << 1), q);
    if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD) {
        const u64 t = PHA_FPSEL(a) ? x : csub(csub(csub(x, q << 2), q << 1), q);
        const u64 r = shoup(sub_mod(aux, t, q), a.scale, q);
        return EPI == EPI_FWD_MODDOWN_ADD ? add_mod(acc, r, q) : r;
    }
    if (EPI == EPI_FWD_KSRESCALE) {   // acc carries the ct word
        const u64 t = PHA_FPSEL(a) ? x : csub(csub(csub(x, q << 2), q << 1), q);
        const u64 u = add_mod(acc, shoup(aux, a.scale2, q), q);
        return shoup(sub_mod(u, t, q), a.scale, q);
    }
    if (EPI == EPI_INV_CANON) return PHA_FPSEL(a) ? x : csub(csub(x, q << 1), q);
    if (EPI == EPI_INV_SCALE) return shoup(x, a.scale, q);
    return x;
}

template <int EPI>
PHA_HD u64 apply_epilogue(u64 x, const PassArgs &a, size_t gi) {
    u64 aux = 0, acc = 0;
    if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE) aux = a.aux[gi];
    if (EPI == EPI_FWD_MODDOWN_ADD) acc = a.out[gi];
    if (EPI == EPI_FWD_KSRESCALE) acc = a.aux2[gi];
    return apply_epilogue_v<EPI>(x, a, aux, acc);
}

// A load that is never served by this CU's vector L1 (agent-scope relaxed: `global_load_dwordx2 ... sc1`, answered
// by the XCD's L2): what the second pass of the one-launch transform reads its intermediate with, because another CU
// of the same XCD wrote it during this launch (MI355X_MICROARCH.md, inter-workgroup visibility).
PHA_HD u64 coherent_load(const u64 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}

// r04 experiment hook (compile-time, default off): cache policy of the coefficient traffic of a pass.  PHA_X_NT bits: 1 = strided pass
// loads, 2 = strided pass stores, 4 = contiguous pass loads, 8 = contiguous pass stores are nontemporal (`nt`: streamed through the
// L2 / MALL without displacing the twiddle rows).  tools/stream_calib.hip: an in-place read-modify-write stream runs at 5.9 TB/s with
// the default policy and at 6.6 TB/s nontemporal on buffers beyond the MALL.
#ifndef PHA_X_NT
#define PHA_X_NT 0
#endif
template <class C, bool LOAD>
constexpr bool x_nt() { return ((PHA_X_NT >> ((C::STRIDED ? 0 : 2) + (LOAD ? 0 : 1))) & 1) != 0; }
#if defined(__clang__)
typedef unsigned long long u64v2 __attribute__((ext_vector_type(2)));
#endif
template <bool NT>
PHA_HD u64 gload(const u64 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <bool NT>
PHA_HD u64x2 gload2(const u64x2 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        const u64v2 t = __builtin_nontemporal_load(reinterpret_cast<const u64v2 *>(p));
        return u64x2{t.x, t.y};
    }
#endif
    return *p;
}
template <bool NT>
PHA_HD void gstore(u64 *p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        __builtin_nontemporal_store(v, p);
        return;
    }
#endif
    *p = v;
}
template <bool NT>
PHA_HD void gstore2(u64x2 *p, u64x2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        __builtin_nontemporal_store(u64v2{v.x, v.y}, reinterpret_cast<u64v2 *>(p));
        return;
    }
#endif
    *p = v;
}

// Load one round's registers: from global memory on the first round, else from LDS.
// COH: the global input was produced by other workgroups of this launch (see coherent_load).
template <class C, int RI, bool FROM_GLOBAL, bool COH = false>
PHA_HD void round_load(const PassArgs &a, const u64 *lds, int tid, u64 *reg) {
    constexpr int r = C::r(RI), K = 1 << r, G = C::EPT >> r, s0 = C::s0(RI);
    constexpr int LOGD = C::LOGT - s0 - r;
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        int v, hi, lo;
        decode_group<C, RI>(tid + C::THREADS * gi, v, hi, lo);
        u64 *rg = reg + gi * K;
        const int e0 = (hi << (LOGD + r)) + lo;
        if (FROM_GLOBAL && COH) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (!C::STRIDED && K == 8 && G == 1 && LOGD >= 1 && LOGD <= 6) {
                // 16-byte L2-served loads (the 8-byte agent-scope form runs at 0.54-0.70x their rate, MI355X_MICROARCH.md):
                // the lanes of a pair (lo = 2m, 2m + 1) need the SAME 16-byte words (k, {2m, 2m + 1}), k = 0..7, one half
                // each; the even lane fetches them for k = 0..3, the odd lane for k = 4..7, and they trade the halves the
                // other one needs with one DPP lane swap per word
                const bool odd = lo & 1;
                const u64 *base = a.in + global_index<C>(a, e0 - (odd ? 1 : 0) + ((odd ? 4 : 0) << LOGD), v);
                u64x2 w0, w1, w2, w3;
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                             "global_load_dwordx4 %1, %4, off offset:%5 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:%6 sc1\n\t"
                             "global_load_dwordx4 %3, %4, off offset:%7 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                             : "v"(base), "n"(8 << LOGD), "n"(16 << LOGD), "n"(24 << LOGD)
                             : "memory");
                const u64x2 w[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u64 send = odd ? w[j].x : w[j].y;
                    const u32 slo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)send, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
                    const u32 shi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(send >> 32), 0xB1, 0xF, 0xF, false);
                    const u64 recv = ((u64)shi << 32) | slo;
                    rg[j] = odd ? recv : w[j].x;          // element (k = j,     lo)
                    rg[4 + j] = odd ? w[j].y : recv;      // element (k = 4 + j, lo)
                }
                continue;
            }
#endif
#pragma unroll
            for (int k = 0; k < K; k++) rg[k] = coherent_load(a.in + global_index<C>(a, e0 + (k << LOGD), v));
        } else if (FROM_GLOBAL) {
            if (!C::STRIDED && LOGD == 0) {  // K contiguous coefficients: 16-byte loads
                const u64x2 *p = reinterpret_cast<const u64x2 *>(a.in + global_index<C>(a, e0, v));
#pragma unroll
                for (int k = 0; k < K; k += 2) {
                    u64x2 t = gload2<x_nt<C, true>()>(p + (k >> 1));
                    rg[k] = t.x;
                    rg[k + 1] = t.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; k++) rg[k] = gload<x_nt<C, true>()>(a.in + global_index<C>(a, e0 + (k << LOGD), v));
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) rg[k] = lds[C::lds_index(e0 + (k << LOGD), v)];
        }
    }
}

// Issue the global loads of one round's twiddles into registers (all rounds are loaded up front,
// together with the data, so a tile pays one memory latency instead of one per round: the tables
// are far larger than L2).
template <class C, int RI>
PHA_HD void round_load_tw(const PassArgs &a, int tid, u64x2 *twreg) {
    constexpr int r = C::r(RI), K = 1 << r, G = C::EPT >> r, s0 = C::s0(RI);
    if (C::ot_round(RI)) {
        // shared factors tw[(hi << j) + kk] (hi is the same for every group of a thread: THREADS is a
        // multiple of 2^s0), then per group the row factors tw[rho << (s0 + j)]
        static_assert(!C::ot_round(RI) || (C::THREADS % (1 << s0)) == 0, "groups of a thread must share hi");
        u64x2 *base = twreg + C::tw_off(RI);
        int v, hi, lo;
        decode_group<C, RI>(tid, v, hi, lo);
        load_group_twiddles<r>(base, a, (u32)hi);
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            decode_group<C, RI>(tid + C::THREADS * gi, v, hi, lo);
            const u32 rho = a.rho0 + a.tile * C::V + (u32)v;
#pragma unroll
            for (int j = 0; j < r; j++) base[(K - 1) + gi * r + j] = tw_entry(a, rho << (s0 + j));
        }
        return;
    }
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        int v, hi, lo;
        decode_group<C, RI>(tid + C::THREADS * gi, v, hi, lo);
        const u32 rho = C::STRIDED ? 1u : (a.rho0 + a.tile * C::V + (u32)v);
#if defined(PHA_EXP_CONST_TW)   // timing experiment only (wrong results): every twiddle from the first 16 entries
        const u32 base0 = 1;
        (void)rho; (void)hi;
#else
        const u32 base0 = (rho << s0) + (u32)hi;
#endif
        load_group_twiddles<r>(twreg + C::tw_off(RI) + gi * (K - 1), a, base0);
    }
}

// Run one round's stages on the registers.
// RECENTRE (forward FP64 path, primes of 47..50 bits): the round runs the LIGHT butterflies and ends by re-centring its registers
// (PassProgram::fp_sched decides after which rounds; pha_arith.h has the magnitude argument).
// GSFULL (inverse FP64 path only): the round is too long for the light butterflies (sums double per stage: more than three stages
// from centred input, more than two from canonical input) and runs the r01 butterflies, which re-centre the sums on their own schedule.
template <class C, int RI, bool FWD, bool FOLD, bool RECENTRE = false, bool GSFULL = false>
PHA_HD void round_compute(const PassArgs &a, int tid, u64 *reg, const u64x2 *twreg) {
    constexpr int r = C::r(RI), K = 1 << r, G = C::EPT >> r;
    const u64 q4 = a.q << 2, nq = 0 - a.q;
    if (C::ot_round(RI)) {
        const u64x2 *tc = twreg + C::tw_off(RI);
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            u64 *rg = reg + gi * K;
            const u64x2 *tr = tc + (K - 1) + gi * r;
            if (PHA_FPSEL(a)) {
                if (FWD ? a.fpm.ct_light : a.fpm.gs_light) ot_round_fp<r, FWD, true>(rg, tc, tr, a.fpm);
                else ot_round_fp<r, FWD, false>(rg, tc, tr, a.fpm);
            } else {
                ot_round_int<r, FWD>(rg, tc, tr, q4, nq);
            }
        }
        return;
    }
    if (PHA_FPSEL(a)) {  // uniform per workgroup
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            u64 *rg = reg + gi * K;
            const u64x2 *t = twreg + C::tw_off(RI) + gi * (K - 1);
            if (FWD) {
                // light butterflies for every prime below 2^50: below 2^47 a whole pass stays exact without any re-centring
                // (ct_light); from 2^47 the registers are re-centred after the rounds fp_sched() names
                fp_ct_round<r>(rg, t, a.fpm);
                if (RECENTRE && !a.fpm.ct_light) {
#pragma unroll
                    for (int k = 0; k < K; k++) rg[k] = as_u64(fp_reduce(as_f64(rg[k]), a.fpm));
                }
            } else if (a.fpm.gs_light) {   // (uniform) primes below 2^42: the r01 rules, no re-centring schedule
                // stages of this pass already done (inverse order: rounds NR-1 .. 0), and whether it began canonical
                constexpr int stage0 = C::LOGT - C::s0(RI) - r;
                constexpr bool canon_in = !C::STRIDED;  // the inverse's first pass is the contiguous one
                // unreduced sums double every stage: a LOGT-stage pass reaches 2^LOGT q, so the whole-transform plans
                // (12 / 13 stages) only run light while that stays within 2^52 (inputs just below q do reach it)
                const bool light = C::LOGT <= 9 || a.fpm.q < (double)(1ull << (52 - C::LOGT));
                if (light) fp_gs_round<r, FOLD, true>(rg, t, a.fpm, a.ninv, a.w1ninv);
                else fp_gs_round<r, FOLD, false, stage0, canon_in>(rg, t, a.fpm, a.ninv, a.w1ninv);
            } else {
                // r04: primes from 2^42: light butterflies (unreduced sums, fp_mulmod_light products) and a re-centring of the registers
                // after the rounds PassProgram::fp_sched() names -- sums double per stage, so at most three stages run between two
                // re-centrings (1/2 -> 1 -> 2 -> 4 q; the product's input |X - Y| stays below 8 q)
                if (GSFULL) {   // from a fresh start: centred input, or canonical input in the first round of the first pass
                    constexpr bool canon = RI == C::NR - 1 && (!C::STRIDED || C::WHOLE);
                    fp_gs_round<r, FOLD, false, 0, canon>(rg, t, a.fpm, a.ninv, a.w1ninv);
                } else {
                    fp_gs_round<r, FOLD, true>(rg, t, a.fpm, a.ninv, a.w1ninv);
                }
                if (RECENTRE) {
#pragma unroll
                    for (int k = 0; k < K; k++) rg[k] = as_u64(fp_reduce(as_f64(rg[k]), a.fpm));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        u64 *rg = reg + gi * K;
        const u64x2 *t = twreg + C::tw_off(RI) + gi * (K - 1);
        if (FWD) ct_round<r>(rg, t, q4, nq);
        else gs_round<r, FOLD>(rg, t, q4, nq, a.ninv, a.w1ninv);
    }
}

// Store one round's registers (to LDS, or to global with the epilogue on the last round).
template <class C, int RI, bool TO_GLOBAL, int EPI>
PHA_HD void round_out(const PassArgs &a, u64 *lds, int tid, const u64 *reg) {
    constexpr int r = C::r(RI), K = 1 << r, G = C::EPT >> r, s0 = C::s0(RI);
    constexpr int LOGD = C::LOGT - s0 - r;
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        int v, hi, lo;
        decode_group<C, RI>(tid + C::THREADS * gi, v, hi, lo);
        const u64 *rg = reg + gi * K;
        const int e0 = (hi << (LOGD + r)) + lo;
        if (TO_GLOBAL) {
            if (!C::STRIDED && LOGD == 0) {
                const size_t g0 = global_index<C>(a, e0, v);
                u64x2 *p = reinterpret_cast<u64x2 *>(a.out + g0);
#pragma unroll
                for (int k = 0; k < K; k += 2) {
                    u64x2 aux{0, 0}, acc{0, 0};
                    if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE)
                        aux = reinterpret_cast<const u64x2 *>(a.aux + g0)[k >> 1];
                    if (EPI == EPI_FWD_MODDOWN_ADD) acc = p[k >> 1];
                    if (EPI == EPI_FWD_KSRESCALE) acc = reinterpret_cast<const u64x2 *>(a.aux2 + g0)[k >> 1];
                    u64x2 t;
                    t.x = apply_epilogue_v<EPI>(rg[k], a, aux.x, acc.x);
                    t.y = apply_epilogue_v<EPI>(rg[k + 1], a, aux.y, acc.y);
                    gstore2<x_nt<C, false>()>(p + (k >> 1), t);
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const size_t gidx = global_index<C>(a, e0 + (k << LOGD), v);
                    gstore<x_nt<C, false>()>(a.out + gidx, apply_epilogue<EPI>(rg[k], a, gidx));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) lds[C::lds_index(e0 + (k << LOGD), v)] = rg[k];
        }
    }
}

// A pass as a list of barrier-separated segments, one per round (a round reads and writes the same
// LDS slots, so only the hand-over between rounds needs a barrier).  FWD runs rounds 0..NR-1,
// inverse NR-1..0.  FOLD (inverse only): this pass holds the transform's last stage (round 0).
// HOIST: 0 = every round requests its own twiddles when it starts; 1 = all rounds up front (one exposed
// memory latency per tile, most registers); 2 = one round ahead (round k+1's twiddles are requested when
// round k starts, so their latency hides behind round k's butterflies and the barrier).
// Hand-over between two rounds: a workgroup barrier, or -- when no wavefront reads another's LDS words --
// only an ordering point for the compiler (LDS operations of one wavefront execute in order).
template <class C>
PHA_HD void tile_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (C::WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
#endif
}

template <class C, bool FWD, int EPI, bool FOLD, int HOIST = 1, bool COH = false>
struct PassProgram {
    static constexpr int NSEG = C::NR;
    static constexpr int THREADS = C::THREADS;
    // the transform's first pass is the strided one going forward and the contiguous one going backward
    static constexpr bool FIRST_PASS = C::WHOLE || (FWD ? C::STRIDED : !C::STRIDED);
    static constexpr bool LAST_PASS = C::WHOLE || !(FWD ? C::STRIDED : !C::STRIDED);

    // FP64 path for primes in [2^47, 2^50) going forward and [2^42, 2^50) going backward (r04): LIGHT butterflies (8 operations instead of 11) plus a re-centring of the
    // registers (3 operations per coefficient) after some rounds.  With |values| <= M q entering a stage, fp_mulmod_light returns
    // |t| <= q (0.5 + 1.5 M q 2^-52) <= q (0.5 + 0.375 M) for q < 2^50, so a stage takes M to 1.375 M + 0.5 (pha_arith.h); every
    // sum and every fma result stays an exact integer while M < 8 (2^53 / 2^50).  fp_sched() follows that bound through the pass
    // in execution order, starting from 1 (canonical input, first pass) or 1/2 (centred input), and re-centres after a round when
    // the next round would pass kFpLimit, and after the last round of a pass that is not the transform's last (so that the
    // intermediate is stored centred and the next pass loads it as it is).  The last pass ends in fp_to_canon, which reduces
    // anything below 2^53.  Rounds with on-the-fly twiddles keep the full butterflies (|t| <= q/2 + 1: M grows by 1/2 per stage).
    // The inverse's sums X + Y double per stage while the products (X - Y) W stay below q (0.5 + 0.75 M): M -> 2 M, so at most three
    // stages from centred input (4 q) or two from canonical input run before a re-centring.
    static constexpr double kFpLimit = 7.5;
    struct FpSched {
        bool recentre[4];
        bool full[4];      // inverse only: the round runs the self-re-centring butterflies (see round_compute GSFULL)
        double after[4];   // bound (units of q) of the registers when round i (execution order) has run, before any re-centring
    };
    static constexpr double fp_grow(double m, int stages, bool full) {
        if (!FWD) {   // inverse: the sum X + Y doubles per stage, the product (X - Y) W comes back below q (0.5 + 0.75 M); rounds with
                      // on-the-fly twiddles and GSFULL rounds re-centre sums themselves and stay below 2 q
            if (full) return 2.0;
            for (int i = 0; i < stages; i++) m = 2.0 * m + 1e-6;
            return m;
        }
        for (int i = 0; i < stages; i++) m = full ? m + 0.5 + 1e-6 : 1.375 * m + 0.5 + 1e-6;
        return m;
    }
    static constexpr FpSched fp_sched() {
        FpSched sc{};
        double m = FIRST_PASS ? 1.0 : 0.5 + 1e-6;
        for (int seg = 0; seg < C::NR; seg++) {
            const int ri = FWD ? seg : C::NR - 1 - seg;
            // (a round that is too long for the light inverse butterflies always starts fresh: from the pass's input, or behind the
            //  re-centring that the previous round's look-ahead asked for)
            sc.full[seg] = !FWD && !C::ot_round(ri) && fp_grow(m, C::r(ri), false) > kFpLimit;
            m = fp_grow(m, C::r(ri), C::ot_round(ri) || sc.full[seg]);
            sc.after[seg] = m;
            bool rc = false;
            if (seg == C::NR - 1) rc = !LAST_PASS;
            else {
                const int rn = FWD ? seg + 1 : C::NR - 2 - seg;
                rc = fp_grow(m, C::r(rn), C::ot_round(rn)) > kFpLimit;
            }
            sc.recentre[seg] = rc;
            if (rc) m = 0.5 + 1e-6;
        }
        return sc;
    }
    static constexpr bool fp_sched_ok() {
        const FpSched sc = fp_sched();
        for (int seg = 0; seg < C::NR; seg++) {
            if (sc.after[seg] > kFpLimit) return false;
            // round_compute returns early for a round with on-the-fly twiddles and ignores RECENTRE / GSFULL there.  Going backward
            // that is sound: the round's full butterflies leave centred values (ot_round_fp reduces every sum), which is what the
            // schedule assumes when it restarts the bound at 1/2 behind it, and `full` is never set for such a round.  Going
            // forward its outputs are NOT centred (M grows by 1/2 per stage), so the schedule must never ask it to re-centre:
            // today it is the last round of the last pass, where the store canonicalises (fp_before_global_store).
            const int ri = FWD ? seg : C::NR - 1 - seg;
            if (C::ot_round(ri) && (sc.full[seg] || (FWD && sc.recentre[seg]))) return false;
        }
        return true;
    }
    static_assert(fp_sched_ok(), "FP64 path: a round would take the registers past 7.5 q (exactness needs < 8 q at 50 bits), or the schedule asks an on-the-fly-twiddle round to re-centre");

    // FP64 path: what just came from global memory becomes a small double.  First pass: canonical integers -> doubles; second
    // pass: the doubles of the first pass -- centred already when the forward pass re-centred them before its store (primes from
    // 2^47), lazy otherwise (the inverse, and the forward's never-re-centred path below 2^47) and then centred here.
    PHA_HD static void fp_after_global_load(const PassArgs &a, u64 *reg) {
        if (!PHA_FPSEL(a)) return;
        if (!FIRST_PASS && !(FWD ? a.fpm.ct_light : a.fpm.gs_light)) return;   // (uniform) the first pass stored centred values
#pragma unroll
        for (int i = 0; i < C::EPT; i++)
            reg[i] = as_u64(FIRST_PASS ? fp_from_canon(reg[i]) : fp_reduce(as_f64(reg[i]), a.fpm));
    }
    // FP64 path, last pass: doubles -> canonical integers (the integer epilogue then sees [0,q))
    PHA_HD static void fp_before_global_store(const PassArgs &a, u64 *reg) {
        if (!PHA_FPSEL(a) || !LAST_PASS) return;
        if (fp_epilogue(EPI)) return;   // the epilogue takes the doubles as they are (apply_epilogue_fp)
#pragma unroll
        for (int i = 0; i < C::EPT; i++) reg[i] = fp_to_canon(as_f64(reg[i]), a.fpm);
    }

    // all twiddle loads of the pass (issued before the first barrier)
    // HOIST: every round's twiddles are requested before the first barrier (one exposed memory latency per
    // tile, more registers); otherwise each round requests its own when it starts (fewer registers).
    static constexpr int round_of(int seg) { return FWD ? seg : C::NR - 1 - seg; }
    PHA_HD static void load_twiddles(const PassArgs &a, int tid, u64x2 *twreg) {
#if defined(PHA_X_NOCOMPUTE) && PHA_X_NOCOMPUTE >= 2
        return;
#endif
        if (HOIST == 1) {
            round_load_tw<C, 0>(a, tid, twreg);
            round_load_tw<C, 1>(a, tid, twreg);
            if (C::NR >= 3) round_load_tw<C, 2>(a, tid, twreg);
            if (C::NR == 4) round_load_tw<C, 3>(a, tid, twreg);
        } else if (HOIST == 2) {
            round_load_tw<C, round_of(0)>(a, tid, twreg);
        }
    }
    template <int SEG>
    PHA_HD static void segment_twiddles(const PassArgs &a, int tid, u64x2 *twreg) {
#if defined(PHA_X_NOCOMPUTE) && PHA_X_NOCOMPUTE >= 2   // 2: no twiddle loads either
        return;
#endif
        if (HOIST == 0) round_load_tw<C, round_of(SEG)>(a, tid, twreg);
        if (HOIST == 2 && SEG + 1 < C::NR) round_load_tw<C, round_of(SEG + 1 < C::NR ? SEG + 1 : SEG)>(a, tid, twreg);
    }

    template <int SEG>
    PHA_HD static void run(const PassArgs &a, u64 *lds, int tid, u64 *reg, u64x2 *twreg) {
        constexpr int RI = FWD ? SEG : C::NR - 1 - SEG;
        constexpr bool first = SEG == 0, last = SEG == C::NR - 1;
        segment_twiddles<SEG>(a, tid, twreg);
        round_load<C, RI, first, COH>(a, lds, tid, reg);
#if defined(__HIP_DEVICE_COMPILE__)
        // r04: ntt_pass_kernel starts every wavefront at raised priority, so that the ~100 scalar / address instructions in front of
        // its global loads are not queued behind the butterflies of the older wavefronts of its SIMD (oldest-first issue): the loads
        // go out at once, the butterflies then run at normal priority (720 limbs -2 %, tools/build_variant.sh prioA)
        if (first) __builtin_amdgcn_s_setprio(0);
#endif
        if (first && FWD && FIRST_PASS && a.pro_reduce) {  // uniform per workgroup
#pragma unroll
            for (int i = 0; i < C::EPT; i++) reg[i] = barrett64(reg[i], a.q, a.pro_ratio1);
        }
        if (first) fp_after_global_load(a, reg);
        // FP64 forward, whole-transform plans beyond 13 stages: magnitudes grow by q/2 + 1 per stage, and the exactness
        // budget of pha_arith.h (|x| < 2^52.6 ~ 6 q at 50 bits going into a multiply) is spent after 13 of them (q + 13 (q/2 + 1)
        // = 7.5 q < 2^52.9 still passes, 14 stages reach 8 q = 2^53), so the registers are re-centred once, after the
        // first 8 stages (3 operations per coefficient)
        if (FWD && C::WHOLE && C::LOGT >= 14 && SEG == 2 && PHA_FPSEL(a)) {
#pragma unroll
            for (int i = 0; i < C::EPT; i++) reg[i] = as_u64(fp_reduce(as_f64(reg[i]), a.fpm));
        }
#if !defined(PHA_X_NOCOMPUTE)   // r04 timing experiment (wrong results): the pass without its butterflies = its memory + LDS floor
        round_compute<C, RI, FWD, FOLD && RI == 0, fp_sched().recentre[SEG], fp_sched().full[SEG]>(a, tid, reg, twreg);
#endif
        if (last) fp_before_global_store(a, reg);
        round_out<C, RI, last, last ? EPI : (int)EPI_NONE>(a, lds, tid, reg);
    }

    // Last segment WITHOUT its store: the registers keep the transform's outputs in the last round's layout (group gi of the
    // thread = K consecutive coefficients starting at global_index(e0, v), see round_out), lazy integers or lazy doubles -- what a
    // fused consumer (the key inner product as the epilogue of the mod-up's contiguous pass) takes over.
    template <int SEG>
    PHA_HD static void run_keep(const PassArgs &a, u64 *lds, int tid, u64 *reg, u64x2 *twreg) {
        constexpr int RI = FWD ? SEG : C::NR - 1 - SEG;
        static_assert(SEG == C::NR - 1 && SEG > 0, "run_keep is the last segment of a multi-round pass");
        segment_twiddles<SEG>(a, tid, twreg);
        round_load<C, RI, false, COH>(a, lds, tid, reg);
        round_compute<C, RI, FWD, FOLD && RI == 0>(a, tid, reg, twreg);
    }

    // Pieces for the software-pipelined (persistent) kernel: the first round's global load is issued
    // one tile ahead of its computation.
    static constexpr int RI_FIRST = FWD ? 0 : C::NR - 1;
    PHA_HD static void prefetch(const PassArgs &a, int tid, u64 *reg) {
        round_load<C, RI_FIRST, true>(a, nullptr, tid, reg);
    }
    template <int SEG>
    PHA_HD static void run_prefetched(const PassArgs &a, u64 *lds, int tid, u64 *reg, u64x2 *twreg) {
        constexpr int RI = FWD ? SEG : C::NR - 1 - SEG;
        constexpr bool first = SEG == 0, last = SEG == C::NR - 1;
        segment_twiddles<SEG>(a, tid, twreg);
        if (!first) round_load<C, RI, false>(a, lds, tid, reg);
        if (first) fp_after_global_load(a, reg);
        round_compute<C, RI, FWD, FOLD && RI == 0, fp_sched().recentre[SEG], fp_sched().full[SEG]>(a, tid, reg, twreg);
        if (last) fp_before_global_store(a, reg);
        round_out<C, RI, last, last ? EPI : (int)EPI_NONE>(a, lds, tid, reg);
    }
};

// Split N = T1 * T2 per log2 N, with the round schedules of each pass.
// VARIANT 0: 16 coefficients per thread (radix-16 rounds, one LDS exchange per pass).
// VARIANT 1: 8 coefficients per thread (radix-8 rounds, two exchanges, 2x the wavefronts).
// VARIANT 2: variant 1 with on-the-fly twiddles in the contiguous pass (see PassCfg::OT).
template <int LOGN, int VARIANT> struct NttPlan;
template <> struct NttPlan<12, 0> { using P1 = PassCfg<6, true, 3, 3>;  using P2 = PassCfg<6, false, 3, 3>; };
template <> struct NttPlan<13, 0> { using P1 = PassCfg<6, true, 3, 3>;  using P2 = PassCfg<7, false, 4, 3>; };
template <> struct NttPlan<14, 0> { using P1 = PassCfg<7, true, 4, 3>;  using P2 = PassCfg<7, false, 4, 3>; };
template <> struct NttPlan<15, 0> { using P1 = PassCfg<7, true, 4, 3>;  using P2 = PassCfg<8, false, 4, 4>; };
template <> struct NttPlan<16, 0> { using P1 = PassCfg<8, true, 4, 4>;  using P2 = PassCfg<8, false, 4, 4>; };
template <> struct NttPlan<17, 0> { using P1 = PassCfg<8, true, 4, 4>;  using P2 = PassCfg<9, false, 3, 3, 3>; };
template <> struct NttPlan<12, 1> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<6, false, 3, 3, 0, 8>; };
template <> struct NttPlan<13, 1> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8>; };
template <> struct NttPlan<14, 1> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8>; };
template <> struct NttPlan<15, 1> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8>; };
template <> struct NttPlan<16, 1> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8>; };
template <> struct NttPlan<17, 1> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8>; };
// VARIANT 2: variant 1 + on-the-fly twiddles in the last round of the contiguous pass
template <> struct NttPlan<12, 2> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<6, false, 3, 3, 0, 8, true>; };
template <> struct NttPlan<13, 2> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, true>; };
template <> struct NttPlan<14, 2> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, true>; };
template <> struct NttPlan<15, 2> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, true>; };
template <> struct NttPlan<16, 2> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, true>; };
template <> struct NttPlan<17, 2> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8, true>; };
// VARIANT 3: variant 1 with one-wavefront workgroups (512-coefficient tiles) in the contiguous pass
template <> struct NttPlan<12, 3> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<6, false, 3, 3, 0, 8, false, 9>; };
template <> struct NttPlan<13, 3> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, false, 9>; };
template <> struct NttPlan<14, 3> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, false, 9>; };
template <> struct NttPlan<15, 3> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, false, 9>; };
template <> struct NttPlan<16, 3> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, false, 9>; };
template <> struct NttPlan<17, 3> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8, false, 9>; };
// VARIANT 4: variant 2 (on-the-fly twiddles) with one-wavefront workgroups in the contiguous pass
template <> struct NttPlan<12, 4> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<6, false, 3, 3, 0, 8, true, 9>; };
template <> struct NttPlan<13, 4> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, true, 9>; };
template <> struct NttPlan<14, 4> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<7, false, 3, 2, 2, 8, true, 9>; };
template <> struct NttPlan<15, 4> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, true, 9>; };
template <> struct NttPlan<16, 4> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<8, false, 3, 3, 2, 8, true, 9>; };
template <> struct NttPlan<17, 4> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8, true, 9>; };
// VARIANT 5 (r03): variant 3 with FOUR coefficients per thread in the contiguous pass -- one T2-point row per wavefront, four
// radix-4 (or 4-4-4-2) rounds: twice the wavefronts of variant 3 with half the serial work each, for launches of a few dozen
// limbs (one co-resident generation: the passes want more, shorter wavefronts).  T2 <= 256 only (four rounds of <= 2 stages).
template <> struct NttPlan<14, 5> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<7, false, 2, 2, 2, 4, false, 8, false, 1>; };
template <> struct NttPlan<15, 5> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<8, false, 2, 2, 2, 4, false, 8, false, 2>; };
template <> struct NttPlan<16, 5> { using P1 = PassCfg<8, true, 3, 3, 2, 8>;  using P2 = PassCfg<8, false, 2, 2, 2, 4, false, 8, false, 2>; };
// VARIANTS 6 / 7 (r03 experiments): variant 5 with four coefficients per thread in the STRIDED pass too -- 6: 2048-coefficient
// tiles (T1 rows x 8 or 16 columns, 512 threads), 7: 4096-coefficient tiles with 1024 threads
template <> struct NttPlan<14, 6> { using P1 = PassCfg<7, true, 2, 2, 2, 4, false, 11, false, 1>;  using P2 = NttPlan<14, 5>::P2; };
template <> struct NttPlan<15, 6> { using P1 = PassCfg<7, true, 2, 2, 2, 4, false, 11, false, 1>;  using P2 = NttPlan<15, 5>::P2; };
template <> struct NttPlan<16, 6> { using P1 = PassCfg<8, true, 2, 2, 2, 4, false, 11, false, 2>;  using P2 = NttPlan<16, 5>::P2; };
template <> struct NttPlan<14, 7> { using P1 = PassCfg<7, true, 2, 2, 2, 4, false, 12, false, 1>;  using P2 = NttPlan<14, 5>::P2; };
template <> struct NttPlan<15, 7> { using P1 = PassCfg<7, true, 2, 2, 2, 4, false, 12, false, 1>;  using P2 = NttPlan<15, 5>::P2; };
template <> struct NttPlan<16, 7> { using P1 = PassCfg<8, true, 2, 2, 2, 4, false, 12, false, 2>;  using P2 = NttPlan<16, 5>::P2; };
// VARIANTS 8 / 9 (r04, experiments library, N = 2^16 only): 2^16 = 128 x 512 instead of 256 x 256 -- the strided pass's tile is 128 rows x 32
// adjacent columns, so its global accesses are runs of 256 bytes instead of 128, and the contiguous pass transforms ONE 512-point
// row per wavefront (three radix-8 rounds).  8: table-driven, 9: on-the-fly twiddles in the last round.
template <> struct NttPlan<16, 8> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8, false, 9>; };
template <> struct NttPlan<16, 9> { using P1 = PassCfg<7, true, 3, 2, 2, 8>;  using P2 = PassCfg<9, false, 3, 3, 3, 8, true, 9>; };
// VARIANTS 10 / 12 (r04; 10 is the product's plan for N = 2^16, 12 an experiment): 2^16 = 64 x 1024 -- strided tile = 64 rows x 64 adjacent columns (512-byte runs, two
// radix-8 rounds, ONE exchange), contiguous pass = 1024-point rows: 10: two wavefronts per row (128 threads x 8, rounds 8-8-4-4),
// 12: one wavefront per row with 16 coefficients per thread (rounds 16-8-8, no workgroup barrier).
// (10's strided tile is 64 rows x 32 columns -- 2048 coefficients, 256 threads: half the barrier width of a 4096-coefficient tile for
//  256-byte runs; measured 1-1.5 % ahead of 64 x 64 tiles on the 720-limb step and on one 45-limb polynomial)
template <> struct NttPlan<16, 10> { using P1 = PassCfg<6, true, 3, 3, 0, 8, false, 11>;  using P2 = PassCfg<10, false, 3, 3, 2, 8, false, 10, false, 2>; };
template <> struct NttPlan<16, 12> { using P1 = PassCfg<6, true, 3, 3, 0, 8>;  using P2 = PassCfg<10, false, 4, 3, 3, 16, false, 10>; };
// VARIANT 13 (r05): plan 10 whose strided pass is the one modup_conv_s1_kernel (pha_ntt.hip) runs from the registers its base conversion
// fills: FOUR coefficients per thread (the conversion's inputs are 30 registers per coefficient), radix-4 rounds 2-2-2, 64 rows x 16
// columns (256 threads); the contiguous pass is plan 10's.  Not a plan choose_plan hands out: the CPU replay (tests/emu) runs it.
template <> struct NttPlan<16, 13> { using P1 = PassCfg<6, true, 2, 2, 2, 4, false, 10>;  using P2 = NttPlan<16, 10>::P2; };
// N = 4096: the whole transform in one 4096-coefficient tile (three radix-16 rounds, one launch, no intermediate)
using WholePlan12 = PassCfg<12, false, 4, 4, 4, 16, false, 12, true>;
// N = 8192: the same with a fourth (radix-2) round: one 8192-coefficient tile, 512 threads, 72 KiB of LDS
using WholePlan13 = PassCfg<13, false, 4, 4, 4, 16, false, 13, true, 1>;
// N = 16384: one 16384-coefficient tile = 128 KiB (+ padding: 144 KiB of gfx950's 160 KiB of LDS), 1024 threads x 16
// coefficients, rounds 16-16-16-4: one workgroup per CU, the intermediate never leaves the chip (fntt_2d.cu:620-653 /
// intt_2d.cu:724-757 at BASELINE config 2's degree)
using WholePlan14 = PassCfg<14, false, 4, 4, 4, 16, false, 14, true, 2>;

}  // namespace pha
