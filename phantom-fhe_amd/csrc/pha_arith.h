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

PHA_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

PHA_HD void mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    lo = a * b;
    hi = mulhi64(a, b);
}

// csub_q (uintmodmath.cuh:18-21)
PHA_HD u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }
PHA_HD u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
PHA_HD u64 sub_mod(u64 a, u64 b, u64 q) { return csub(a + q - b, q); }
PHA_HD u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// multiply_and_reduce_shoup_lazy (uintmodmath.cuh:223-231): any 64-bit a, result in [0,2q)
PHA_HD u64 shoup_lazy(u64 a, u64x2 w, u64 q) { return a * w.x - mulhi64(a, w.y) * q; }
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
    u64 quo = hi * m.ratio1 + tmp3 + carry;
    return csub(lo - quo * m.value, m.value);
}
// multiply_and_barrett_reduce_uint64 (:160-198)
PHA_HD u64 mul_mod(u64 a, u64 b, const DModulus &m) {
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return barrett128(lo, hi, m);
}
// barrett_reduce_uint64_uint64 (:144-151)
PHA_HD u64 barrett64(u64 x, u64 q, u64 ratio1) { return csub(x - mulhi64(x, ratio1) * q, q); }

// 128-bit accumulate helper
PHA_HD void mac128(u64 a, u64 b, u64 &lo, u64 &hi) {
    u64 pl, ph;
    mul128(a, b, pl, ph);
    lo += pl;
    hi += ph + (lo < pl);
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
