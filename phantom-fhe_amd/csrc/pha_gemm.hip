// pha_gemm.hip -- batched modular GEMM: C[z] = A[z] * B[z] mod q_z for a batch of RNS moduli.
//
// Reference: benchmark/matmul_bench.cu:215-541 (gemm_v00 .. v04; 256^3 per modulus, 30 moduli of 50 bits), row-major
// A [m][k], B [k][n], C [m][n] per modulus, 128-bit accumulate and one Barrett reduction per output.  NOTE: the
// reference's kernels add the low and the high product words separately (`sum.lo += a * b; sum.hi += __umul64hi(a, b)`,
// :231-232) and so lose the carries out of the low word; only its all-ones benchmark input hides that.  This kernel
// computes the exact product modulo q (for the reference's own input the two agree).
//
// Still 64-bit integer arithmetic: no MFMA.  LDS-tiled (64 x 32 outputs per workgroup, 16 deep, 2 x 4 outputs per thread), operands cut
// once, while they are staged, into SB-bit halves; every multiply-accumulate is four carry-free v_mad_u64_u32
// into plain 64-bit partial sums (the trick of the base converter).  SB = 25 for moduli below 2^50 (partial
// products below 2^50: the sums never overflow for k <= 16384, one recombination at the very end); SB = 30 for
// moduli up to 2^60 (partial products below 2^60: sixteen terms per partial sum, so the partial sums are folded
// into 128-bit totals after every 16-deep tile, and the totals are Barrett-reduced to one word every 8 tiles so that
// they cannot overflow for any k).
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

namespace pha {

constexpr int kGemmBM = 64, kGemmBN = 32, kGemmBK = 16, kGemmThreads = 256;
constexpr int kGemmTM = 2, kGemmTN = 4;   // outputs per thread: 2 rows x 4 columns -> 32 x 8 threads

struct GemmArgs {
    u64 *c;
    const u64 *a, *b;
    const DModulus *mod;     // modulus of batch entry z is mod[mod_start + z]
    uint32_t mod_start, m, n, k;
    size_t lda, ldb, ldc;
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
    GemmArgs g{C, A, B, c.d_mod.p, (uint32_t)mod_start_idx, (uint32_t)m, (uint32_t)n, (uint32_t)k, lda, ldb, ldc};
    const dim3 grid((unsigned)((n + kGemmBN - 1) / kGemmBN), (unsigned)((m + kGemmBM - 1) / kGemmBM), (unsigned)batch);
    if (narrow) hipLaunchKernelGGL((gemm_mod_kernel<25, false>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
    else hipLaunchKernelGGL((gemm_mod_kernel<30, true>), grid, dim3(kGemmThreads), 0, as_stream(stream), g);
    check_launch();
    PHA_API_END
}
