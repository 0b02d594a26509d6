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
// one wavefront per SIMD.  B fragments (a lane = a column, 16 consecutive k) come straight from coalesced loads; A
// (a lane = a row) is loaded coalesced along k, cut into digits, and transposed through 7-8 KB of LDS.  The integer
// multiply-add version this replaced (carry-free v_mad_u64_u32 on 25/30-bit halves, LDS-tiled) ran at 68 % of the
// v_mad_u64_u32 peak and is kept below under PHA_GEMM_VALU for comparison builds (profiles/r03_experiments.md).
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

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

// exact value of sum_s acc_s 256^s (acc_s signed 32-bit, the total is non-negative) as a 128-bit number
template <int NACC>
__device__ __forceinline__ void recombine(const int (&a)[NACC], u64 &lo, u64 &hi) {
    unsigned __int128 tot = 0;
#pragma unroll
    for (int g = 0; g * 4 < NACC; g++) {
        long long part = 0;      // four digit weights at a time: |part| < 2^31 * 2^24 * 2
#pragma unroll
        for (int s = 3; s >= 0; s--)
            if (g * 4 + s < NACC) part = part * 256 + (long long)a[g * 4 + s];
        tot += (unsigned __int128)(__int128)part << (32 * g);
    }
    lo = (u64)tot;
    hi = (u64)(tot >> 64);
}

// One wavefront = one 32 x 32 tile of C for one modulus.  FULL: m, n, k multiples of 32 (no bounds checks).
// ACCUM: add to what C holds (the launcher cuts k into runs of 224 for the wide moduli, where 256 q^2 can pass 2^128).
template <int D, bool FULL, bool ACCUM>
__global__ __launch_bounds__(64, 1) void gemm_mfma_kernel(const GemmArgs g, uint32_t tiles_m, uint32_t tiles_n) {
    constexpr int NACC = 2 * D - 1;
    __shared__ u32 lds[D * 256];          // D planes x 32 rows x 32 bytes of k
    const uint32_t lane = threadIdx.x;
    // workgroups are handed to the 8 XCDs in turn: give each XCD a contiguous run of tiles (neighbours share A rows / a modulus in its L2)
    const uint32_t total = gridDim.x, w = blockIdx.x;
    const uint32_t t = (total % 8 == 0) ? (w % 8) * (total / 8) + w / 8 : w;
    const uint32_t nb = t % tiles_n, mb = (t / tiles_n) % tiles_m, z = t / (tiles_n * tiles_m);
    const DModulus mo = g.mod[g.mod_start + z];
    const u64 *A = g.a + (size_t)z * g.a_batch_stride;
    const u64 *B = g.b + (size_t)z * g.b_batch_stride;
    u64 *C = g.c + (size_t)z * g.m * g.ldc;
    const uint32_t row0 = mb * 32, col0 = nb * 32;
    const uint32_t arow = lane >> 3, akq = (lane & 7) * 4;        // A loads: 8 lanes x 4 operands cover the 32 k of a row, 8 rows per pass
    const uint32_t bj = lane & 31, bg = (lane >> 5) * 16;         // B loads = B fragment: column bj, k = bg .. bg + 15

    v16i acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; s++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[s][r] = 0;
    u64 out[16];

    u64 ra[4][4], rb[16];
    auto load_raw = [&](uint32_t k0) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t gr = row0 + p * 8 + arow;
            const u64 *src = A + (size_t)gr * g.lda + k0 + akq;
#pragma unroll
            for (int e = 0; e < 4; e++) ra[p][e] = (FULL || (gr < g.m && k0 + akq + e < g.k)) ? src[e] : 0;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint32_t gk = k0 + bg + e, gc = col0 + bj;
            rb[e] = (FULL || (gk < g.k && gc < g.n)) ? B[(size_t)gk * g.ldb + gc] : 0;
        }
    };
    const uint32_t nkb = (g.k + 31) / 32;
    auto block = [&](uint32_t kb) {
        // digits of A -> LDS (plane d, row, 4 bytes of k per lane); digits of B stay in registers as the B fragments
        v4i bfrag[D];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            u32 pl[8];
            planes_of4<D>(ra[p], pl);
#pragma unroll
            for (int d = 0; d < D; d++) lds[d * 256 + p * 64 + lane] = pl[d];      // (p * 8 + arow) * 8 + (lane & 7) = p * 64 + lane
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 x4[4] = {rb[q * 4], rb[q * 4 + 1], rb[q * 4 + 2], rb[q * 4 + 3]};
            u32 pl[8];
            planes_of4<D>(x4, pl);
#pragma unroll
            for (int d = 0; d < D; d++) bfrag[d][q] = (int)pl[d];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) load_raw((kb + 1) * 32);      // in flight under the MFMAs of this block
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        v4i afrag[D];
#pragma unroll
        for (int d = 0; d < D; d++) afrag[d] = *reinterpret_cast<const v4i *>(&lds[d * 256 + (lane & 31) * 8 + (lane >> 5) * 4]);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < D; j++) acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[i], bfrag[j], acc[i + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_raw(0);
    for (uint32_t kb = 0; kb < nkb; kb++) block(kb);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        int a[NACC];
#pragma unroll
        for (int s = 0; s < NACC; s++) a[s] = acc[s][r];
        u64 lo, hi;
        recombine<NACC>(a, lo, hi);
        out[r] = barrett128(lo, hi, mo);
        __builtin_amdgcn_sched_barrier(0);      // one output at a time: keeps the 13-15 accumulator reads of the others out of the VGPRs
    }
    // C/D layout of the 32 x 32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gc = col0 + (lane & 31);
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
    const uint32_t tm = (uint32_t)((m + 31) / 32), tn = (uint32_t)((n + 31) / 32);
    const size_t tiles = (size_t)tm * tn * batch;
    if (tiles > 0x7fffffffull) throw std::invalid_argument("gemm shape is not valid");
    const bool full = m % 32 == 0 && n % 32 == 0 && k % 32 == 0;
    const dim3 grid((unsigned)tiles), block(64);
    // narrow moduli: 7 digits, the int32 accumulators hold k <= 16384 (7 * 16384 * 2^14 < 2^31) and the 128-bit total k q^2 < 2^114;
    // wide: 8 digits, k in runs of 224 (224 q^2 < 2^128), every run after the first added to C
    auto launch = [&](auto kernel, const GemmArgs &ga) { hipLaunchKernelGGL(kernel, grid, block, 0, as_stream(stream), ga, tm, tn); };
    if (narrow) {
        if (full) launch(gemm_mfma_kernel<7, true, false>, g);
        else launch(gemm_mfma_kernel<7, false, false>, g);
    } else {
        for (size_t k0 = 0; k0 < k; k0 += 224) {
            GemmArgs ga = g;
            ga.a = A + k0;
            ga.b = B + k0 * ldb;
            ga.k = (uint32_t)std::min<size_t>(224, k - k0);
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
