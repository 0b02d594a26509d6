// pha_arith.h -- 64-bit modular arithmetic for gfx950 (CDNA4).
//
// Semantics follow the reference's device primitives (include/uintmodmath.cuh:18-242,
// include/butterfly.cuh:10-37) but the code is written for the AMD VALU: there is no native
// 64x64 multiplier, so every product is a short chain of v_mad_u64_u32 / v_mul_lo_u32 /
// v_mul_hi_u32, and 64-bit adds are single v_lshl_add_u64.  All functions are also valid host
// C++ so that tests/emu can replay a kernel's thread program on the CPU (test-only; the product
// never executes them on the host).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHA_HD __host__ __device__ __forceinline__
#else
#define PHA_HD inline
#endif

// The FP64 residue arithmetic below (and every kernel that includes this header) relies on separately rounded
// multiplies and adds: h = Y*W must round before fma(Y, W, -h) recovers its error term.  The rule is stated here, in
// the source, so that it does not depend on the build's -ffp-contract flag (a CXXFLAGS override used to drop it):
// at file scope the pragma holds to the end of the translation unit, i.e. for everything that includes this header.
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

namespace pha {

typedef uint64_t u64;
typedef unsigned int u32;

// (value, Shoup companion floor(value * 2^64 / q)) -- one 16-byte load fetches both.
struct alignas(16) u64x2 {
    u64 x, y;
};

// DModulus (include/ntt.cuh:6-32): value + Barrett ratio floor(2^128/q) as two words.
struct DModulus {
    u64 value;
    u64 ratio0;
    u64 ratio1;
};

// v_mad_u64_u32 helpers.  v_mul_hi_u32 issues at 1/8 of the FP32 rate on gfx950, v_mad_u64_u32 / v_mul_lo_u32 at 1/4, and the mad
// delivers the full 64-bit product (profiles/r01_microbench_gfx950.txt), so every 64 x 64 product below is built from them
// (the compiler's own expansion of __umul64hi and of a 64-bit `a * b` goes through v_mul_hi_u32).
PHA_HD u64 mad_u64_u32(u32 a, u32 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (u64)a * b + c;
#endif
}
PHA_HD u64 mul_u64_u32(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
#else
    return (u64)a * b;
#endif
}
PHA_HD u64 add_u64_u32(u64 c, u32 a) {  // c + a without a carry chain through VCC
#if defined(__HIP_DEVICE_COMPILE__)
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(carry) : "v"(a), "v"(c));
    return d;
#else
    return c + a;
#endif
}
// exact floor(a*b / 2^64) and a*b mod 2^64
PHA_HD u64 mulhi64_mad(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = mul_u64_u32(a0, b0);
    const u64 p01 = mad_u64_u32(a0, b1, p00 >> 32);     // (2^32-1)^2 + 2^32-1 < 2^64
    const u64 p10 = mad_u64_u32(a1, b0, (u32)p01);
    return mad_u64_u32(a1, b1, p01 >> 32) + (p10 >> 32);
}
PHA_HD u64 mullo64_mad(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    return mul_u64_u32(a0, b0) + ((u64)(u32)(a0 * b1 + a1 * b0) << 32);
}

PHA_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mulhi64_mad(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

PHA_HD u64 mullo64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mullo64_mad(a, b);
#else
    return a * b;
#endif
}

PHA_HD void mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    lo = mullo64(a, b);
    hi = mulhi64(a, b);
}

// csub_q (uintmodmath.cuh:18-21)
PHA_HD u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }
PHA_HD u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
PHA_HD u64 sub_mod(u64 a, u64 b, u64 q) { return csub(a + q - b, q); }
PHA_HD u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// multiply_and_reduce_shoup_lazy (uintmodmath.cuh:223-231): any 64-bit a, result in [0,2q)
PHA_HD u64 shoup_lazy(u64 a, u64x2 w, u64 q) { return mullo64(a, w.x) - mullo64(mulhi64(a, w.y), q); }
// multiply_and_reduce_shoup (:207-215): canonical
PHA_HD u64 shoup(u64 a, u64x2 w, u64 q) { return csub(shoup_lazy(a, w, q), q); }

// barrett_reduce_uint128_uint64 (uintmodmath.cuh:96-136): (hi:lo) mod q, canonical.
PHA_HD u64 barrett128(u64 lo, u64 hi, const DModulus &m) {
    u64 carry = mulhi64(lo, m.ratio0);
    u64 t_lo, t_hi;
    mul128(lo, m.ratio1, t_lo, t_hi);
    u64 tmp1 = t_lo + carry;
    u64 tmp3 = t_hi + (tmp1 < t_lo);
    mul128(hi, m.ratio0, t_lo, t_hi);
    u64 s = tmp1 + t_lo;
    carry = t_hi + (s < tmp1);
    u64 quo = mullo64(hi, m.ratio1) + tmp3 + carry;
    return csub(lo - mullo64(quo, m.value), m.value);
}
// multiply_and_barrett_reduce_uint64 (:160-198)
PHA_HD u64 mul_mod(u64 a, u64 b, const DModulus &m) {
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return barrett128(lo, hi, m);
}
// barrett_reduce_uint64_uint64 (:144-151)
PHA_HD u64 barrett64(u64 x, u64 q, u64 ratio1) { return csub(x - mullo64(mulhi64(x, ratio1), q), q); }

// 128-bit accumulate helper
PHA_HD void mac128(u64 a, u64 b, u64 &lo, u64 &hi) {
    u64 pl, ph;
    mul128(a, b, pl, ph);
    lo += pl;
    hi += ph + (lo < pl);
}

// ---- gfx950-tuned lazy Shoup multiply ---------------------------------------------------------------
// v_mul_hi_u32 issues at 1/8 of the FP32 rate on gfx950, v_mad_u64_u32 at 1/4 and delivers the full
// 64-bit product (profiles/r01_microbench_gfx950.txt), so the quotient estimate is built from three
// v_mad_u64_u32 (the compiler would pick v_mul_hi_u32, hence the asm), drops the a0*b0 term and the
// middle carry (estimate in [Q-2, Q]), and the remainder Y*w - Q*q is one v_mad chain against -q.
// Result in [0, 4q); any 64-bit Y.  Measured 87.7 vs 106 cycles per wave-butterfly.
// floor(a*b / 2^64) - e, e in {0,1,2}
PHA_HD u64 mulhi64_approx(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p01 = mul_u64_u32(a0, b1), p10 = mul_u64_u32(a1, b0);
    return add_u64_u32(mad_u64_u32(a1, b1, p01 >> 32), (u32)(p10 >> 32));
}
// Y*w mod q, lazy: result in [0,4q). nq = 2^64 - q.
PHA_HD u64 shoup_lazy4(u64 Y, u64x2 w, u64 nq) {
    const u32 y0 = (u32)Y, y1 = (u32)(Y >> 32), w0 = (u32)w.x, w1 = (u32)(w.x >> 32);
    const u64 Q = mulhi64_approx(Y, w.y);
    const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 T = mad_u64_u32(q0, n0, mul_u64_u32(y0, w0));
    const u32 hi = (u32)(T >> 32) + y0 * w1 + y1 * w0 + q0 * n1 + q1 * n0;
    return ((u64)hi << 32) | (u32)T;
}

// Montgomery reduction: (hi:lo) * 2^-64 mod p, canonical, for odd p, (hi:lo) < 2^64 * p and ninv = -p^-1 mod 2^64.
// m = lo * ninv makes (hi:lo) + m * p divisible by 2^64; the quotient is hi + floor(m * p / 2^64) + (lo != 0) < 2p.
PHA_HD u64 mont_redc128(u64 lo, u64 hi, u64 p, u64 ninv) {
    const u64 m = mullo64_mad(lo, ninv);
    const u64 t = hi + mulhi64_mad(m, p) + (lo != 0 ? 1 : 0);
    return csub(t, p);
}

// r06: Montgomery reduction of the four carry-free split accumulators of a base conversion, word by word in base B = 2^30, without
// first recombining them into a 128-bit value (22 vector instructions) for the 64-bit REDC above (18):
//   V = ll + (lh + hl) B + hh B^2      ->      V * B^-3 mod p, canonical,
// for odd p < 2^60 cut as p = p0 + p1 B, n30 = -p^-1 mod 2^30 (any upper bits of n30 are ignored), and accumulators of AT MOST 15
// products of 30-bit halves: with H = 2^30 - 1 each of ll, lh, hl, hh is <= 15 H^2, and every sum below stays inside 64 bits:
//   step 1  m1 = ll n30 mod B;  t = ll + m1 p0 <= 16 H^2 = 2^64 - 2^35 + 16 (low 30 bits zero);  u = lh + m1 p1 + (t >> 30)
//           <= 16 H^2 + 2^34 - 32 < 2^64;                                   V / B   = u + hl + hh B
//   step 2  m2 = (u + hl) n30 mod B;  w = hl + m2 p0 <= 16 H^2;  u + w < 2^65 (its carry is kept);  x = hh + m2 p1 + ((u + w) >> 30)
//           <= 16 H^2 + 2^35 - 48 = 2^64 - 32;                               V / B^2 = x
//   step 3  m3 = x n30 mod B;  r = (x + m3 p) >> 30 = (x >> 30) + m3 p1 + ((x mod B + m3 p0) >> 30) < 2^34 + 2^60 + 2^30
// r = (V + (m1 + m2 B + m3 B^2) p) / B^3 < V / 2^90 + p, and V < 15 * 2^60 * p for rows below p, so r < 2p: one conditional subtract.
// The rows of such a converter hold qhat_i * 2^90 mod p_j (BConv::r90, pha_context.hip).  A sixteenth term would overflow step 1:
// converters with 16 inputs keep the 64-bit form.  Checked against big-integer arithmetic at the corners by tests/test_emu_fp.py.
PHA_HD u64 mont_redc90_split(u64 ll, u64 lh, u64 hl, u64 hh, u64 p, u32 p0, u32 p1, u32 n30) {
    constexpr u32 M = (1u << 30) - 1;
    const u32 m1 = ((u32)ll * n30) & M;
    const u64 t = mad_u64_u32(m1, p0, ll);
    const u64 u = mad_u64_u32(m1, p1, lh) + (t >> 30);
    const u32 m2 = (((u32)u + (u32)hl) * n30) & M;
    const u64 w = mad_u64_u32(m2, p0, hl);
    const u64 s = u + w;
    const u64 c = s < u ? (u64)1 << 34 : 0;
    const u64 x = mad_u64_u32(m2, p1, hh) + ((s >> 30) | c);
    const u32 x0 = (u32)x & M;
    const u32 m3 = (x0 * n30) & M;
    const u64 r = mad_u64_u32(m3, p1, x >> 30) + (mad_u64_u32(m3, p0, (u64)x0) >> 30);
    return csub(r, p);
}

// Butterflies on the [0,8q) / [0,4q) lazy ranges that shoup_lazy4 needs (q < 2^61). q4 = 4q.
// CT: X,Y in [0,8q) -> [0,8q)
PHA_HD void ct_bfly4(u64 &X, u64 &Y, u64x2 w, u64 q4, u64 nq) {
    const u64 x = csub(X, q4);
    const u64 t = shoup_lazy4(Y, w, nq);
    X = x + t;
    Y = x + q4 - t;
}
// GS: X,Y in [0,4q) -> [0,4q)
PHA_HD void gs_bfly4(u64 &X, u64 &Y, u64x2 w, u64 q4, u64 nq) {
    const u64 s = X + Y;
    const u64 d = X + q4 - Y;
    X = csub(s, q4);
    Y = shoup_lazy4(d, w, nq);
}

// ---- FP64 residue arithmetic for primes q < 2^50 ------------------------------------------------------
// On gfx950 v_fma_f64 / v_mul_f64 / v_add_f64 issue at the same rate as v_mad_u64_u32, and a modular
// multiply needs 9 of them with no carry chains or zero-extensions (52.8 vs 87.7 cycles per
// wave-butterfly, profiles/r01_microbench_gfx950.txt).  Residues are held as doubles with integer values,
// |x| < 2^52.6, so sums/differences are exact; products are exact through the fma error term:
//   h = Y*W (rounded), l = fma(Y, W, -h) = Y*W - h exactly, c = rint(h * fl(1/q)) is the quotient to
//   within 3 (three roundings of relative size 2^-53 on a value below 2^52.6), r = fma(-c, q, h) + l =
//   Y*W - c*q exactly (an integer below 2^52), and one more rint/fma step centres it:
//   result == Y*W (mod q), |result| <= q/2 + 1.  The twiddle tables of this path therefore hold W alone
//   (8 bytes per entry; the integer path needs 16 for W and its Shoup quotient).
// Every stored output is converted back to the canonical integer residue, so results are bit-identical
// to the integer path.
PHA_HD double as_f64(u64 x) { return __builtin_bit_cast(double, x); }
PHA_HD u64 as_u64(double x) { return __builtin_bit_cast(u64, x); }
struct FpMod {
    double q, qinv;  // q and fl(1/q)
    // Light butterflies (no re-centring inside a pass) are exact while every magnitude stays below 2^52.6:
    //   forward: |t| <= q (0.5 + 1.5 |Y| 2^-52) per stage, 8 stages per pass -> q < 2^47 keeps |x| < 11 q < 2^50.5
    //   inverse: sums double per stage, 8 stages per pass                  -> q < 2^43 suffices (|x| < 2^8 q)
    // (a pass = at most 9 stages for N = 2^17; the thresholds below keep one more bit of margin for that).
    // The one-launch N = 4096 transform runs 12 stages in its single pass: forward light still fits (|x| < 10 q <
    // 2^50.3), the full forward butterfly reaches q + 12 (q/2 + 1) < 7.1 q < 2^52.9 at q < 2^50 (integers below
    // 2^53 are exact and fp_mulmod's quotient stays within 3 there), and the inverse only runs light below 2^40
    // (round_compute in pha_ntt_core.h).
    // r04: primes from 2^47 up to 2^50 run the forward transform on the light butterflies too, with the registers re-centred
    // after the rounds PassProgram::fp_sched() (pha_ntt_core.h) names: with |values| <= M q entering a stage, |t| <= q (0.5 + 0.375 M)
    // at q < 2^50, i.e. M -> 1.375 M + 0.5 per stage (1/2 -> 1.19 -> 2.13 -> 3.43 -> 5.22 from centred, 1 -> 1.88 -> 3.08 -> 4.73 ->
    // 7.0 from canonical input); everything stays an exact integer while M < 8.  ct_light (below 2^47) marks the primes that
    // need no re-centring inside a pass at all.
    bool ct_light, gs_light;
};
PHA_HD FpMod make_fpmod(u64 q) { return FpMod{(double)q, 1.0 / (double)q, (q >> 47) == 0, (q >> 42) == 0}; }
// x - rint(x/q)*q : |result| <= q/2 + 1 for |x| < 2^52.6
PHA_HD double fp_reduce(double x, FpMod m) { return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x); }
// Y*W mod q, centred. W in [0,q), |Y| < 2^52.6
PHA_HD double fp_mulmod(double Y, double W, FpMod m) {
    const double h = Y * W;
    const double l = __builtin_fma(Y, W, -h);
    const double c = __builtin_rint(h * m.qinv);
    const double r = __builtin_fma(-c, m.q, h) + l;
    return fp_reduce(r, m);
}
// Y*W - c*q with c = rint(h * fl(1/q)): == Y*W (mod q), |result| <= q (0.5 + 1.5 |Y| 2^-52): already centred
// when |Y| << 2^52, so small primes skip fp_mulmod's second step (3 of its 10 operations, all on the dependent
// chain)
PHA_HD double fp_mulmod_light(double Y, double W, FpMod m) {
    const double h = Y * W;
    const double l = __builtin_fma(Y, W, -h);
    const double c = __builtin_rint(h * m.qinv);
    return __builtin_fma(-c, m.q, h) + l;
}
// GS butterfly: (X, Y) -> ((X + Y) mod q, (X - Y)*W mod q), both centred (inputs |.| <= 2q)
PHA_HD void fp_gs_bfly(double &X, double &Y, double W, FpMod m) {
    const double s = X + Y, d = X - Y;
    X = fp_reduce(s, m);
    Y = fp_mulmod(d, W, m);
}
// canonical integer residue [0,q) (q < 2^50) <-> double, via the 2^52 mantissa trick
PHA_HD double fp_from_canon(u64 x) { return as_f64(x | 0x4330000000000000ull) - 4503599627370496.0; }
PHA_HD u64 fp_to_canon(double x, FpMod m) {
    double r = fp_reduce(x, m);              // [-q/2-1, q/2+1]
    r = r < 0.0 ? r + m.q : r;               // [0, q)
    return as_u64(r + 4503599627370496.0) & 0x000fffffffffffffull;
}
// r04: the 2 x 2 tensor product / square of a limb below 2^50 on canonical residues (tensor_prod_2x2_rns_poly polymath.cu:463-496,
// tensor_square_2x2_rns_poly :498-529): three light products.  The unreduced sums c0 + c1 are below 2 q, so the quotient estimate
// of their product is off by at most 0.5 + 1.5 * 2 * 2 q 2^-52 <= 2 and its light reduction stays below 2 q + 2^48; with d0 and d2
// below 0.875 q each, d1 is an exact integer below 3.75 q + 2^48 when fp_to_canon reduces it (tests/test_emu_fp.py).
PHA_HD void fp_tensor_2x2(u64 a0, u64 a1, u64 b0, u64 b1, FpMod m, u64 &d0, u64 &d1, u64 &d2) {
    const double x0 = fp_from_canon(a0), x1 = fp_from_canon(a1), y0 = fp_from_canon(b0), y1 = fp_from_canon(b1);
    const double e0 = fp_mulmod_light(x0, y0, m), e2 = fp_mulmod_light(x1, y1, m);
    const double e1 = fp_mulmod_light(x0 + x1, y0 + y1, m) - e0 - e2;
    d0 = fp_to_canon(e0, m);
    d1 = fp_to_canon(e1, m);
    d2 = fp_to_canon(e2, m);
}
PHA_HD void fp_square_2x2(u64 a0, u64 a1, FpMod m, u64 &d0, u64 &d1, u64 &d2) {
    const double x0 = fp_from_canon(a0), x1 = fp_from_canon(a1);
    d0 = fp_to_canon(fp_mulmod_light(x0, x0, m), m);
    d1 = fp_to_canon(fp_mulmod_light(x0 + x0, x1, m), m);
    d2 = fp_to_canon(fp_mulmod_light(x1, x1, m), m);
}

// Harvey butterflies (include/butterfly.cuh:10-22 / :28-37). q2 = 2q.
// CT: X,Y in [0,4q) -> X,Y in [0,4q)
PHA_HD void ct_bfly(u64 &X, u64 &Y, u64x2 w, u64 q, u64 q2) {
    u64 x = csub(X, q2);
    u64 t = shoup_lazy(Y, w, q);
    X = x + t;
    Y = x + q2 - t;
}
// GS: X,Y in [0,2q) -> X,Y in [0,2q)
PHA_HD void gs_bfly(u64 &X, u64 &Y, u64x2 w, u64 q, u64 q2) {
    u64 s = X + Y;
    u64 d = X + q2 - Y;
    X = csub(s, q2);
    Y = shoup_lazy(d, w, q);
}

}  // namespace pha
