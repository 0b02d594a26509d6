// pha_gemm.hip -- batched modular GEMM: C[z] = A[z] * B[z] mod q_z for a batch of RNS moduli.
//
// Reference: benchmark/matmul_bench.cu:215-541 (gemm_v00 .. v04; 256^3 per modulus, 30 moduli of 50 bits), row-major
// A [m][k], B [k][n], C [m][n] per modulus, 128-bit accumulate and one Barrett reduction per output.  NOTE: the
// reference's kernels add the low and the high product words separately (`sum.lo += a * b; sum.hi += __umul64hi(a, b)`,
// :231-232) and so lose the carries out of the low word; only its all-ones benchmark input hides that.  This kernel
// computes the exact product modulo q (for the reference's own input the two agree).
//
// This row IS a GEMM, so it runs on the matrix cores: every 64-bit operand is cut into D signed bytes
// (balanced base-256 digits: the bytes of (x + 0x80..80) ^ 0x80..80, D = 7 for moduli below 2^50, 8 up to 2^60), the
// D x D digit-plane products go through v_mfma_i32_32x32x32_i8 into 2 D - 1 int32 accumulators (one per digit weight
// 256^(i+j)), and the exact 128-bit sum is put back together and Barrett-reduced once per output (once per 128 of k for the
// wide moduli, where k q^2 can pass 2^128).  One wavefront owns a 32 x 32 tile of C: 13 (15) accumulators x 16 registers,
// one wavefront per SIMD; four of them form a workgroup that shares the loads and the digit cutting of a 64 x 64 tile through
// LDS (layouts and pipeline at the kernel).  Algorithmic work: 49 (64) i8 multiply-adds per modular one, so 30 x 256^3 is
// 24.7 G i8 MACs = 11 us at the measured 4.4 POP/s i8 rate; the kernel runs at about 30 % of that (profiles/r03_experiments.md
// has the phase stamps: MFMA issue is 63 % of the main loop, the reduction of the accumulators a quarter of a tile's time).
// The integer multiply-add version this replaced (carry-free v_mad_u64_u32 on 25/30-bit halves, LDS-tiled) ran at 68 % of
// the v_mad_u64_u32 peak (101 us) and is kept below under PHA_GEMM_VALU for comparison builds.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

#ifndef PHA_GEMM_VALU_PER_MFMA
#define PHA_GEMM_VALU_PER_MFMA 4      // vector instructions the scheduler places after each MFMA (3: slower, 4-5: same, 8: slower)
#endif
#ifndef PHA_GEMM_X
#define PHA_GEMM_X 0      // 5: phase stamps instead of the result (profiles/HISTORY.md 4.9)
#endif

namespace pha {

#if PHA_GEMM_VALU
constexpr int kGemmBM = 64, kGemmBN = 32, kGemmBK = 16, kGemmThreads = 256;
constexpr int kGemmTM = 2, kGemmTN = 4;   // outputs per thread: 2 rows x 4 columns -> 32 x 8 threads
#endif

struct GemmArgs {
    u64 *c;
    const u64 *a, *b;
    const DModulus *mod;     // modulus of batch entry z is mod[mod_start + z]
    uint32_t mod_start, m, n, k;
    size_t lda, ldb, ldc;
    size_t a_batch_stride, b_batch_stride;   // operands per modulus (m lda, k ldb of the whole problem: k may be one run of it)
};

// value = ll + (lh + hl) * 2^SB + hh * 2^(2 SB) as a 128-bit number added to (lo, hi)
template <int SB>
__device__ __forceinline__ void fold(u64 ll, u64 lh, u64 hl, u64 hh, u64 &lo, u64 &hi) {
    const u64 mid = lh + hl;
    const u64 mid_c = mid < lh ? 1 : 0;                // mid may carry into bit 64 (SB = 30)
    u64 t = mid << SB;
    u64 h = (mid >> (64 - SB)) + (mid_c << SB);
    lo += ll;
    hi += lo < ll;
    lo += t;
    hi += (lo < t) + h;
    t = hh << (2 * SB);
    h = hh >> (64 - 2 * SB);
    lo += t;
    hi += (lo < t) + h;
}

#if PHA_GEMM_VALU
template <int SB, bool FOLD_EACH_TILE>
__global__ __launch_bounds__(kGemmThreads) void gemm_mod_kernel(const GemmArgs g) {
    __shared__ uint2 As[kGemmBK][kGemmBM + 1];
    __shared__ uint2 Bs[kGemmBK][kGemmBN];
    const uint32_t z = blockIdx.z;
    const DModulus mo = g.mod[g.mod_start + z];
    const u64 *A = g.a + (size_t)z * g.m * g.lda;
    const u64 *B = g.b + (size_t)z * g.k * g.ldb;
    u64 *C = g.c + (size_t)z * g.m * g.ldc;
    const uint32_t tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;      // 8 column groups x 32 row groups
    const uint32_t row0 = blockIdx.y * kGemmBM, col0 = blockIdx.x * kGemmBN;
    constexpr u32 MASK = (1u << SB) - 1;

    u64 acc[kGemmTM][kGemmTN][4];
    u64 tot_lo[kGemmTM][kGemmTN], tot_hi[kGemmTM][kGemmTN];
#pragma unroll
    for (int i = 0; i < kGemmTM; i++)
#pragma unroll
        for (int j = 0; j < kGemmTN; j++) {
            acc[i][j][0] = acc[i][j][1] = acc[i][j][2] = acc[i][j][3] = 0;
            tot_lo[i][j] = tot_hi[i][j] = 0;
        }

    uint32_t tiles_since_reduce = 0;
    for (uint32_t k0 = 0; k0 < g.k; k0 += kGemmBK) {
        // stage A (64 rows x 16: four coefficients per thread) and B (16 x 32: two per thread), cut into halves on the way
        {
            const uint32_t r = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t gr = row0 + r, gk = k0 + kk + e;
                const u64 v = (gr < g.m && gk < g.k) ? A[(size_t)gr * g.lda + gk] : 0;
                As[kk + e][r] = uint2{(u32)v & MASK, (u32)(v >> SB)};
            }
            const uint32_t kr = tid >> 4, cc = (tid & 15) * 2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t gk = k0 + kr, gc = col0 + cc + e;
                const u64 v = (gk < g.k && gc < g.n) ? B[(size_t)gk * g.ldb + gc] : 0;
                Bs[kr][cc + e] = uint2{(u32)v & MASK, (u32)(v >> SB)};
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kGemmBK; kk++) {
            uint2 av[kGemmTM], bv[kGemmTN];
#pragma unroll
            for (int i = 0; i < kGemmTM; i++) av[i] = As[kk][ty * kGemmTM + i];
#pragma unroll
            for (int j = 0; j < kGemmTN; j++) bv[j] = Bs[kk][tx * kGemmTN + j];
#pragma unroll
            for (int i = 0; i < kGemmTM; i++)
#pragma unroll
                for (int j = 0; j < kGemmTN; j++) {
                    acc[i][j][0] = (u64)av[i].x * bv[j].x + acc[i][j][0];
                    acc[i][j][1] = (u64)av[i].x * bv[j].y + acc[i][j][1];
                    acc[i][j][2] = (u64)av[i].y * bv[j].x + acc[i][j][2];
                    acc[i][j][3] = (u64)av[i].y * bv[j].y + acc[i][j][3];
                }
        }
        __syncthreads();
        if (FOLD_EACH_TILE) {
#pragma unroll
            for (int i = 0; i < kGemmTM; i++)
#pragma unroll
                for (int j = 0; j < kGemmTN; j++) {
                    fold<SB>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], tot_lo[i][j], tot_hi[i][j]);
                    acc[i][j][0] = acc[i][j][1] = acc[i][j][2] = acc[i][j][3] = 0;
                }
            // products of two 60-bit residues are below 2^120: the 128-bit totals hold 128 of them plus one reduced
            // word, so they go back to a single word every 8 tiles (k = 128); without this k > 256 would overflow
            if (++tiles_since_reduce == 8) {
                tiles_since_reduce = 0;
#pragma unroll
                for (int i = 0; i < kGemmTM; i++)
#pragma unroll
                    for (int j = 0; j < kGemmTN; j++) {
                        tot_lo[i][j] = barrett128(tot_lo[i][j], tot_hi[i][j], mo);
                        tot_hi[i][j] = 0;
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kGemmTM; i++) {
        const uint32_t gr = row0 + ty * kGemmTM + i;
#pragma unroll
        for (int j = 0; j < kGemmTN; j++) {
            const uint32_t gc = col0 + tx * kGemmTN + j;
            if (!FOLD_EACH_TILE) fold<SB>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], tot_lo[i][j], tot_hi[i][j]);
            if (gr < g.m && gc < g.n) C[(size_t)gr * g.ldc + gc] = barrett128(tot_lo[i][j], tot_hi[i][j], mo);
        }
    }
}

#endif  // PHA_GEMM_VALU

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// bytes d of four 32-bit words -> four words holding byte 0, 1, 2, 3 of (x0, x1, x2, x3) (v_perm_b32: selector 0-3 = bytes
// of the SECOND operand, 4-7 = bytes of the first)
__device__ __forceinline__ void byte_transpose4(u32 x0, u32 x1, u32 x2, u32 x3, u32 &p0, u32 &p1, u32 &p2, u32 &p3) {
    const u32 t01l = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t01h = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
    const u32 t23l = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t23h = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
    p0 = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
    p1 = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
    p2 = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
    p3 = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
}

// D signed bytes of x (x = sum_d s_d 256^d, s_d in [-128, 127]): the bytes of (x + OFFS) ^ OFFS, OFFS = 0x80 in the low D - 1 bytes
template <int D>
__device__ __forceinline__ u64 balanced_bytes(u64 x) {
    constexpr u64 OFFS = D == 7 ? 0x0000808080808080ull : 0x0080808080808080ull;
    return (x + OFFS) ^ OFFS;
}

// D digit planes (4 bytes each) of four operands
template <int D>
__device__ __forceinline__ void planes_of4(const u64 (&x)[4], u32 (&pl)[8]) {
    u64 y[4];
#pragma unroll
    for (int e = 0; e < 4; e++) y[e] = balanced_bytes<D>(x[e]);
    byte_transpose4((u32)y[0], (u32)y[1], (u32)y[2], (u32)y[3], pl[0], pl[1], pl[2], pl[3]);
    byte_transpose4((u32)(y[0] >> 32), (u32)(y[1] >> 32), (u32)(y[2] >> 32), (u32)(y[3] >> 32), pl[4], pl[5], pl[6], pl[7]);
}

// a * b + c on signed operands: one v_mad_i64_i32 (b uniform: the digit weights 2^8, 2^16, 2^24)
__device__ __forceinline__ long long mad_i64_i32(int a, int b, long long c) {
    long long d;
    u64 carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    return d;
}

// exact value of sum_s acc_s 256^s (acc_s signed 32-bit, the total is non-negative and below 2^128) as a 128-bit number:
// four digit weights at a time into a signed 64-bit group g_i (|g_i| < 2^56), total = g0 + g1 2^32 + g2 2^64 + g3 2^96
template <int NACC>
__device__ __forceinline__ void recombine(const int (&a)[NACC], u64 &lo, u64 &hi) {
    long long grp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        long long v = i * 4 < NACC ? (long long)a[i * 4] : 0;
        if (i * 4 + 1 < NACC) v = mad_i64_i32(a[i * 4 + 1], 1 << 8, v);
        if (i * 4 + 2 < NACC) v = mad_i64_i32(a[i * 4 + 2], 1 << 16, v);
        if (i * 4 + 3 < NACC) v = mad_i64_i32(a[i * 4 + 3], 1 << 24, v);
        grp[i] = v;
    }
    lo = (u64)grp[0] + ((u64)grp[1] << 32);
    const long long carry = lo < (u64)grp[0] ? 1 : 0;
    hi = (u64)((grp[0] >> 63) + (grp[1] >> 32) + carry) + (u64)grp[2] + ((u64)grp[3] << 32);
}

// 32 x 32 -> 64 product, b uniform (an SGPR: no copy of the modulus constants into vector registers)
__device__ __forceinline__ u64 mul_u64_u32_s(u32 a, u32 b) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    return d;
}
// floor(a b / 2^64) or one less (the a0 b0 product is left out), b uniform
__device__ __forceinline__ u64 mulhi64_near(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p01 = mul_u64_u32_s(a0, b1), p10 = mul_u64_u32_s(a1, b0), p11 = mul_u64_u32_s(a1, b1);
    const u64 mid = (u64)(u32)p01 + (u32)p10;
    return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}
__device__ __forceinline__ u64 mullo64_s(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    return mul_u64_u32_s(a0, b0) + ((u64)(u32)(a0 * b1 + a1 * b0) << 32);
}
// (hi : lo) mod q for q < 2^60 with ratio = floor(2^128 / q): the quotient estimate hi r1 + hi(hi r0) + hi(lo r1) is at most 6
// short (three dropped fractions, the floor in ratio, two near-products), so the remainder is below 7 q < 2^63
__device__ __forceinline__ u64 reduce128_uniform(u64 lo, u64 hi, const DModulus &m) {
    const u64 quo = mullo64_s(hi, m.ratio1) + mulhi64_near(hi, m.ratio0) + mulhi64_near(lo, m.ratio1);
    u64 r = lo - mullo64_s(quo, m.value);
    r -= r >= 4 * m.value ? 4 * m.value : 0;
    r -= r >= 2 * m.value ? 2 * m.value : 0;
    r -= r >= m.value ? m.value : 0;
    return r;
}

// One workgroup of four wavefronts (one per SIMD) = one 64 x 64 tile of C for one modulus, each wavefront a 32 x 32 quarter.
// FULL: m, n multiples of 64 and k of 64 (no bounds checks, an even number of 32-deep blocks).
// ACCUM: add to what C holds (the launcher cuts k into runs of 128 for the wide moduli, where 256 q^2 can pass 2^128).
//
// Per 32-deep block of k the workgroup loads 64 x 32 of A and 32 x 64 of B once (8 + 8 operands per thread, coalesced), cuts them
// into digit planes and leaves those in LDS:
//   A plane d: [row 0..63][32 bytes of k]               -> a fragment (row = lane & 31, 16 k) is one ds_read_b128
//   B plane d: [k octet 0..3][column 0..63][8 bytes of k] -> a fragment (column = lane & 31, octets 2 g, 2 g + 1) is two ds_read_b64;
//   a thread holds eight consecutive k of one row / column, so every plane is one ds_write_b64 at 8 * thread: no bank conflicts
// Two LDS buffers and two operand register sets: in step kb the loads of block kb + 2 go out, the vector ALU cuts block kb + 1
// (loaded during step kb - 1) into the other LDS buffer while the D^2 MFMAs of block kb run on the matrix pipe, then one
// barrier and the fragments of block kb + 1 are read.  Measured and not kept (profiles/r03_experiments.md): the fragment reads
// under the MFMAs as well (a second fragment set), loads three steps ahead (four operand sets).
constexpr int kGemmWg = 256;

template <int D, bool FULL, bool ACCUM>
__global__ __launch_bounds__(kGemmWg, 1) void gemm_mfma_kernel(const GemmArgs g, uint32_t tiles_m, uint32_t tiles_n) {
    constexpr int NACC = 2 * D - 1;
    constexpr int PLANE = 512;                           // 4-byte words per plane (2 KB)
    __shared__ u32 lds[2][2][D * PLANE];                 // [buffer][A / B][plane][...]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wm = wave & 1, wn = wave >> 1;
    // workgroups are handed to the 8 XCDs in turn: give each XCD a contiguous run of tiles (neighbours share A rows / a modulus in its L2)
    const uint32_t total = gridDim.x, w = blockIdx.x;
    const uint32_t t = (total % 8 == 0) ? (w % 8) * (total / 8) + w / 8 : w;
    const uint32_t nb = t % tiles_n, mb = (t / tiles_n) % tiles_m, z = t / (tiles_n * tiles_m);
    const DModulus mo = g.mod[g.mod_start + z];
    const u64 *A = g.a + (size_t)z * g.a_batch_stride;
    const u64 *B = g.b + (size_t)z * g.b_batch_stride;
    u64 *C = g.c + (size_t)z * g.m * g.ldc;
    const uint32_t row0 = mb * 64, col0 = nb * 64;
    const uint32_t arow = tid >> 2, ak8 = (tid & 3) * 8;       // A loads: 4 threads x 8 operands cover the 32 k of a row
    const uint32_t bj = tid & 63, bk8 = (tid >> 6) * 8;         // B loads: column bj, k = bk8 .. bk8 + 7

    v16i acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; s++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[s][r] = 0;
    u64 out[16];

    // loads: uniform (scalar) tile bases + loop-invariant 32-bit byte offsets per thread (the launcher bounds lda, ldb)
    const char *Atile = reinterpret_cast<const char *>(A + (size_t)row0 * g.lda);
    const char *Btile = reinterpret_cast<const char *>(B + col0);
    const uint32_t aoff = (uint32_t)((arow * g.lda + ak8) * 8);
    const uint32_t boff = (uint32_t)((bk8 * g.ldb + bj) * 8);
    struct Raw {
        u64 a[2][4], b[2][4];             // two runs of four consecutive k each
    };
    Raw r0, r1;                           // operands of two blocks in flight
    auto load_raw = [&](Raw &raw, uint32_t k0) {
        const char *ab = Atile + (size_t)k0 * 8;
        const uint32_t gr = row0 + arow;
#pragma unroll
        for (int e = 0; e < 8; e++)
            raw.a[e >> 2][e & 3] = (FULL || (gr < g.m && k0 + ak8 + e < g.k)) ? *reinterpret_cast<const u64 *>(ab + aoff + 8 * e) : 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const char *bb = Btile + ((size_t)k0 + e) * g.ldb * 8;             // + bk8 rows through boff
            raw.b[e >> 2][e & 3] = (FULL || (k0 + bk8 + e < g.k && col0 + bj < g.n)) ? *reinterpret_cast<const u64 *>(bb + boff) : 0;
        }
    };
    // raw operands -> digit planes in LDS buffer `buf`: eight bytes (eight consecutive k) per thread and plane, one ds_write_b64
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    auto decompose = [&](const Raw &raw, uint32_t buf) {
        u32 *la = lds[buf][0], *lb = lds[buf][1];
        u32 p0[8], p1[8];
        planes_of4<D>(raw.a[0], p0);
        planes_of4<D>(raw.a[1], p1);
#pragma unroll
        for (int d = 0; d < D; d++) *reinterpret_cast<u32x2 *>(&la[d * PLANE + tid * 2]) = u32x2{p0[d], p1[d]};    // row arow, bytes ak8 .. ak8 + 7
        planes_of4<D>(raw.b[0], p0);
        planes_of4<D>(raw.b[1], p1);
#pragma unroll
        for (int d = 0; d < D; d++) *reinterpret_cast<u32x2 *>(&lb[d * PLANE + tid * 2]) = u32x2{p0[d], p1[d]};    // [k octet][column]
    };
    v4i afrag[D], bfrag[D];
    auto take_frags = [&](uint32_t buf) {
        __syncthreads();
        const u32 *la = lds[buf][0] + (wm * 32 + (lane & 31)) * 8 + (lane >> 5) * 4;
        const u32 *lb = lds[buf][1] + (lane >> 5) * 256 + (wn * 32 + (lane & 31)) * 2;
#pragma unroll
        for (int d = 0; d < D; d++) {
            afrag[d] = *reinterpret_cast<const v4i *>(la + d * PLANE);
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(lb + d * PLANE), hi = *reinterpret_cast<const u32x2 *>(lb + d * PLANE + 128);
            bfrag[d] = v4i{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
        }
    };
    // FULL means an even number of blocks; otherwise a block past the end loads zeros and adds nothing.
    const uint32_t nkb = (g.k + 31) / 32;
    auto block_k0 = [&](uint32_t b) { return (FULL ? min(b, nkb - 1) : b) * 32; };
    auto step = [&](Raw &ld, const Raw &dc, uint32_t kb, uint32_t nextbuf) {
        load_raw(ld, block_k0(kb + 2));
        decompose(dc, nextbuf);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < D; j++) acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[i], bfrag[j], acc[i + j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 12, 0);          // issue order: the loads,
#pragma unroll
        for (int i = 0; i < D * D; i++) {                            // then one MFMA and the vector work that hides under it
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, PHA_GEMM_VALU_PER_MFMA, 0);
        }
        take_frags(nextbuf);
        __builtin_amdgcn_sched_barrier(0);                           // the two steps of a pair are scheduled separately
    };
#if PHA_GEMM_X == 5      // phase stamps (profiles/HISTORY.md 4.9): s_memtime over C instead of the result
    u64 stamp[8];
    int ns = 0;
#define PHA_STAMP() do { __builtin_amdgcn_sched_barrier(0); if (ns < 8) stamp[ns++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PHA_STAMP() do { } while (0)
#endif
    PHA_STAMP();
    load_raw(r0, 0);
    decompose(r0, 0);
    load_raw(r1, block_k0(1));
    take_frags(0);
    PHA_STAMP();
    for (uint32_t kb = 0; kb < nkb; kb += 2) {
        step(r0, r1, kb, 1);
        step(r1, r0, kb + 1, 0);
        PHA_STAMP();
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        int a[NACC];
#pragma unroll
        for (int s = 0; s < NACC; s++) a[s] = acc[s][r];
        u64 lo, hi;
        recombine<NACC>(a, lo, hi);
        out[r] = reduce128_uniform(lo, hi, mo);
        __builtin_amdgcn_sched_barrier(0);      // one output at a time: keeps the 13-15 accumulator reads of the others out of the VGPRs
    }
#if PHA_GEMM_X == 5
    u64 sink = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) sink ^= out[r];
    asm volatile("" :: "v"(sink));
    PHA_STAMP();
    if (tid == 0)
        for (int i = 0; i < 8; i++) g.c[(size_t)t * 8 + i] = stamp[i];
    return;
#endif
    // C/D layout of the 32 x 32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gc = col0 + wn * 32 + (lane & 31);
        if (FULL || (gr < g.m && gc < g.n)) {
            u64 v = out[r];
            if (ACCUM) {
                v += C[(size_t)gr * g.ldc + gc];
                v -= v >= mo.value ? mo.value : 0;
            }
            C[(size_t)gr * g.ldc + gc] = v;
        }
    }
}

}  // namespace pha

using namespace pha;

extern "C" int pha_batched_modular_gemm(pha_context_t ctx, uint64_t *C, size_t ldc, const uint64_t *A, size_t lda,
                                        const uint64_t *B, size_t ldb, size_t m, size_t n, size_t k, size_t batch,
                                        size_t mod_start_idx, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !C || !A || !B) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (batch == 0 || m == 0 || n == 0) return 0;
    if (mod_start_idx + batch > c.rows) throw std::invalid_argument("modulus index out of range");
    if (batch > 65535 || k == 0 || k > 16384 || lda < k || ldb < n || ldc < n) throw std::invalid_argument("gemm shape is not valid");
    if ((lda | ldb | ldc) >> 23) throw std::invalid_argument("gemm leading dimension exceeds 2^23");     // 32-bit lane offsets in the kernel
    bool narrow = true;   // every modulus below 2^50: 25-bit halves, no folding inside the k loop
    for (size_t z = 0; z < batch; z++) {
        const u64 q = c.primes[mod_start_idx + z];
        if (q >> 60) throw std::invalid_argument("modulus exceeds 60 bits");
        if (q >> 50) narrow = false;
    }
    GemmArgs g{C, A, B, c.d_mod.p, (uint32_t)mod_start_idx, (uint32_t)m, (uint32_t)n, (uint32_t)k, lda, ldb, ldc, m * lda, k * ldb};
#if PHA_GEMM_VALU
    const dim3 grid((unsigned)((n + kGemmBN - 1) / kGemmBN), (unsigned)((m + kGemmBM - 1) / kGemmBM), (unsigned)batch);
    if (narrow) hipLaunchKernelGGL((gemm_mod_kernel<25, false>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
    else hipLaunchKernelGGL((gemm_mod_kernel<30, true>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
#else
    const uint32_t tm = (uint32_t)((m + 63) / 64), tn = (uint32_t)((n + 63) / 64);
    const size_t tiles = (size_t)tm * tn * batch;
    if (tiles > 0x7fffffffull) throw std::invalid_argument("gemm shape is not valid");
    const bool full = m % 64 == 0 && n % 64 == 0 && k % 64 == 0;      // whole tiles and an even number of 32-deep blocks
    const dim3 grid((unsigned)tiles), block(kGemmWg);
    // narrow moduli: 7 digits, the int32 accumulators hold k <= 16384 (7 * 16384 * 2^14 < 2^31) and the 128-bit total k q^2 < 2^114;
    // wide: 8 digits, k in runs of 128 (128 q^2 < 2^127; four blocks), every run after the first added to C
    auto launch = [&](auto kernel, const GemmArgs &ga) { hipLaunchKernelGGL(kernel, grid, block, 0, as_stream(stream), ga, tm, tn); };
    if (narrow) {
        if (full) launch(gemm_mfma_kernel<7, true, false>, g);
        else launch(gemm_mfma_kernel<7, false, false>, g);
    } else {
        for (size_t k0 = 0; k0 < k; k0 += 128) {
            GemmArgs ga = g;
            ga.a = A + k0;
            ga.b = B + k0 * ldb;
            ga.k = (uint32_t)std::min<size_t>(128, k - k0);
            ga.a_batch_stride = m * lda;
            ga.b_batch_stride = k * ldb;
            if (k0 == 0) {
                if (full) launch(gemm_mfma_kernel<8, true, false>, ga);
                else launch(gemm_mfma_kernel<8, false, false>, ga);
            } else {
                if (full) launch(gemm_mfma_kernel<8, true, true>, ga);
                else launch(gemm_mfma_kernel<8, false, true>, ga);
            }
        }
    }
#endif
    check_launch();
    PHA_API_END
}
