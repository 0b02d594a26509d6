// emu_fp.cpp -- TEST-ONLY: the FP64 modular helpers of phantom-fhe_amd/csrc/pha_arith.h compiled for the host (they are
// host/device functions), so that their exactness claims are checked against big-integer arithmetic without a GPU
// (tests/test_emu_fp.py).  Built with -ffp-contract=off like the library.
#include <cstddef>
#include <cstdint>
#include "../../phantom-fhe_amd/csrc/pha_arith.h"

using namespace pha;

extern "C" {

// out[3 i ..] = tensor product (d0, d1, d2) of (a0[i], a1[i]) x (b0[i], b1[i]) modulo q (fp_tensor_2x2)
void emu_fp_tensor(uint64_t q, size_t count, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1, uint64_t *out) {
    const FpMod m = make_fpmod(q);
    for (size_t i = 0; i < count; i++) {
        u64 d0, d1, d2;
        fp_tensor_2x2(a0[i], a1[i], b0[i], b1[i], m, d0, d1, d2);
        out[3 * i] = d0; out[3 * i + 1] = d1; out[3 * i + 2] = d2;
    }
}
void emu_fp_square(uint64_t q, size_t count, const uint64_t *a0, const uint64_t *a1, uint64_t *out) {
    const FpMod m = make_fpmod(q);
    for (size_t i = 0; i < count; i++) {
        u64 d0, d1, d2;
        fp_square_2x2(a0[i], a1[i], m, d0, d1, d2);
        out[3 * i] = d0; out[3 * i + 1] = d1; out[3 * i + 2] = d2;
    }
}
// out[i] = canonical (y[i] * w[i]) mod q through fp_mulmod (full) or fp_mulmod_light; y is given as a signed multiple-of-q offset:
// the product is formed on Y = y[i] + shift * q (|Y| < 2^52.6 is the caller's business), W = w[i] in [0, q)
void emu_fp_mulmod(uint64_t q, size_t count, const uint64_t *y, const uint64_t *w, int shift, int light, uint64_t *out) {
    const FpMod m = make_fpmod(q);
    for (size_t i = 0; i < count; i++) {
        const double Y = fp_from_canon(y[i]) + (double)shift * m.q, W = fp_from_canon(w[i]);
        out[i] = fp_to_canon(light ? fp_mulmod_light(Y, W, m) : fp_mulmod(Y, W, m), m);
    }
}
// largest |fp_mulmod_light(Y, W)| / q over the inputs, times 2^20 (the bound the re-centring schedules build on)
uint64_t emu_fp_light_bound(uint64_t q, size_t count, const uint64_t *y, const uint64_t *w, int shift) {
    const FpMod m = make_fpmod(q);
    double worst = 0.0;
    for (size_t i = 0; i < count; i++) {
        const double Y = fp_from_canon(y[i]) + (double)shift * m.q, W = fp_from_canon(w[i]);
        const double r = fp_mulmod_light(Y, W, m);
        const double a = r < 0 ? -r : r;
        worst = a > worst ? a : worst;
    }
    return (uint64_t)(worst / m.q * 1048576.0);
}
}

// r06: out[i] = mont_redc90_split(ll[i], lh[i], hl[i], hh[i]) for the modulus p (the word-wise Montgomery reduction of the base
// conversion's split accumulators): compared with (ll + (lh + hl) 2^30 + hh 2^60) * 2^-90 mod p in Python
extern "C" void emu_redc90(uint64_t p, size_t count, const uint64_t *ll, const uint64_t *lh, const uint64_t *hl, const uint64_t *hh, uint64_t *out) {
    uint64_t inv = p;   // Newton: p * inv = 1 mod 2^64
    for (int it = 0; it < 6; it++) inv *= 2 - p * inv;
    const u32 n30 = (u32)(0 - inv);   // -p^-1 mod 2^32: the function masks it to 30 bits
    for (size_t i = 0; i < count; i++)
        out[i] = mont_redc90_split(ll[i], lh[i], hl[i], hh[i], p, (u32)(p & 0x3fffffffu), (u32)(p >> 30), n30);
}
// the accumulators the conversion kernels form for inputs y[term][i] and one Montgomery row m[term] (30 / 30 cuts)
extern "C" void emu_split_acc(size_t terms, size_t count, const uint64_t *y, const uint64_t *m, uint64_t *ll, uint64_t *lh, uint64_t *hl, uint64_t *hh) {
    for (size_t i = 0; i < count; i++) {
        u64 a = 0, b = 0, c = 0, d = 0;
        for (size_t t = 0; t < terms; t++) {
            const u32 y0 = (u32)y[t * count + i] & 0x3fffffffu, y1 = (u32)(y[t * count + i] >> 30);
            const u32 m0 = (u32)m[t] & 0x3fffffffu, m1 = (u32)(m[t] >> 30);
            a += (u64)y0 * m0; b += (u64)y0 * m1; c += (u64)y1 * m0; d += (u64)y1 * m1;
        }
        ll[i] = a; lh[i] = b; hl[i] = c; hh[i] = d;
    }
}
