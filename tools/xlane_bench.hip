// xlane_bench.hip -- is a cross-lane exchange (ds_swizzle / DPP-style butterflies, "wavefront shuffles at the inner radices")
// cheaper than the LDS round trip between two rounds of the one-wavefront contiguous NTT pass?
//
// Between two radix-8 rounds a thread must trade 8 x u64 registers with 7 other lanes: an 8 x 8 transpose between the register
// index and three lane-index bits (lane bits 2..4 for the exchange after the first round of a 256-point row).  Two forms:
//   lds     : what the pass does -- 8 ds_write_b64 into the padded tile, 8 ds_read_b64 back in the new order (no barrier: one
//             wavefront per tile);
//   swizzle : three butterfly steps; in step s a lane keeps half of its registers and trades the other half with lane ^ (4 << s)
//             through ds_swizzle_b32 (bit mode xor mask; no LDS memory is touched), selecting with v_cndmask.
// Both are timed inside a loop with a data dependence between iterations, at 1..8 wavefronts per SIMD, plus a multiply-add
// filler so that the VALU is as busy as in the real pass.  Build: hipcc -O3 --offload-arch=gfx950 tools/xlane_bench.hip -o tools/xlane_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned long long u64;

template <int XOR>
__device__ __forceinline__ unsigned swz(unsigned v) {
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (XOR << 10) | 0x1f);   // bit mode: and 0x1f, or 0, xor XOR
}
template <int S>
__device__ __forceinline__ void butterfly_step(u64 (&v)[8], bool b) {
#pragma unroll
    for (int r0 = 0; r0 < 8; r0++) {
        if (r0 & (1 << S)) continue;
        const int r1 = r0 | (1 << S);
        const u64 send = b ? v[r0] : v[r1];
        const unsigned lo = swz<(4 << S)>((unsigned)send), hi = swz<(4 << S)>((unsigned)(send >> 32));
        const u64 recv = ((u64)hi << 32) | lo;
        if (b) v[r0] = recv;
        else v[r1] = recv;
    }
}

// r04: the same 8 x 8 transpose with the gfx950 VALU cross-lane instructions, which do NOT go through the LDS
// pipe: register bit 2 <-> lane bit 5 by v_permlane32_swap (swaps the upper 32 lanes of one register with the lower 32 of another:
// exactly one transposition step per register pair, one instruction per dword pair), register bit 1 <-> lane bit 4 by
// v_permlane16_swap (odd rows of one register <-> even rows of the other), register bit 0 <-> lane bit 3 by DPP row_ror:8 moves
// with a bank mask (lane i <- lane i ^ 8 inside a row of 16; two masked moves per dword pair).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int WHICH>
__device__ __forceinline__ void swap_pair(unsigned &a, unsigned &b) {   // a = register with the bit clear, b = with the bit set
    if (WHICH == 2) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r.x;
        b = r.y;
    } else if (WHICH == 1) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r.x;
        b = r.y;
    } else {
        // lanes with bit 3 clear (banks 0, 1 of each row) take b <- a of lane ^ 8; lanes with bit 3 set (banks 2, 3) take a <- b of lane ^ 8
        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x128 /* row_ror:8 */, 0xf, 0x3, false);
        const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x128, 0xf, 0xc, false);
        a = na;
        b = nb;
    }
}
template <int BIT>
__device__ __forceinline__ void permlane_step(u64 (&v)[8]) {
#pragma unroll
    for (int r0 = 0; r0 < 8; r0++) {
        if (r0 & (1 << BIT)) continue;
        const int r1 = r0 | (1 << BIT);
        unsigned a0 = (unsigned)v[r0], a1 = (unsigned)(v[r0] >> 32), b0 = (unsigned)v[r1], b1 = (unsigned)(v[r1] >> 32);
        swap_pair<BIT>(a0, b0);
        swap_pair<BIT>(a1, b1);
        v[r0] = ((u64)a1 << 32) | a0;
        v[r1] = ((u64)b1 << 32) | b0;
    }
}

template <int MODE, int FILL>
__global__ __launch_bounds__(64) void exch_kernel(u64 *out, int iters) {
    __shared__ u64 tile[2 * (256 + 32)];
    const int lane = threadIdx.x, row = lane >> 5, lo = lane & 31;
    u64 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (u64)lane * 8 + k + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {   // LDS: write e = lo + 32 k, read e = hi' * 32 + 4 k + lo' (the pass's index maps, padded by 2 words per 16)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = lo + 32 * k;
                tile[row * 288 + e + ((e >> 4) << 1)] = v[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int hi2 = lo >> 2, lo2 = lo & 3;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = hi2 * 32 + 4 * k + lo2;
                v[k] = tile[row * 288 + e + ((e >> 4) << 1)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else if (MODE == 1) {   // swizzle butterflies over lane bits 2, 3, 4
            butterfly_step<0>(v, (lane >> 2) & 1);
            butterfly_step<1>(v, (lane >> 3) & 1);
            butterfly_step<2>(v, (lane >> 4) & 1);
        } else if (MODE == 3) {   // VALU cross-lane transposes over lane bits 3, 4, 5
            permlane_step<0>(v);
            permlane_step<1>(v);
            permlane_step<2>(v);
        }
#pragma unroll
        for (int f = 0; f < FILL; f++)   // VALU filler with the cost profile of the butterflies (v_fma_f64 rate)
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = __builtin_bit_cast(u64, __builtin_fma(__builtin_bit_cast(double, v[k] | 0x3ff0000000000000ull), 1.0000001, 0.5));
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] += it;
    }
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc ^= v[k];
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int FILL>
static double run(int waves_per_simd, int iters) {
    const int blocks = 256 * 4 * waves_per_simd;
    u64 *d;
    hipMalloc(&d, (size_t)blocks * 64 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    exch_kernel<MODE, FILL><<<blocks, 64>>>(d, 16);
    hipEventRecord(a);
    exch_kernel<MODE, FILL><<<blocks, 64>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms * 1e-3 / iters * 2.4e9 / waves_per_simd;   // cycles (at 2.4 GHz) per exchange per wavefront, per SIMD
}

int main() {
    printf("cycles per exchange per wavefront (nominal 2.4 GHz), 8 x u64 registers <-> lane bits 2..4; wavefronts per SIMD 1 / 2 / 4 / 8\n");
    const int w[4] = {1, 2, 4, 8};
    printf("%-34s", "LDS round trip, no filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<0, 0>(w[i], 4096));
    printf("\n%-34s", "ds_swizzle butterflies, no filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<1, 0>(w[i], 4096));
    printf("\n%-34s", "permlane swap + DPP, no filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<3, 0>(w[i], 4096));
    printf("\n%-34s", "filler only (12 x 8 fma_f64)");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<2, 12>(w[i], 4096));
    printf("\n%-34s", "LDS round trip + filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<0, 12>(w[i], 4096));
    printf("\n%-34s", "ds_swizzle butterflies + filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<1, 12>(w[i], 4096));
    printf("\n%-34s", "permlane swap + DPP + filler");
    for (int i = 0; i < 4; i++) printf(" %8.1f", run<3, 12>(w[i], 4096));
    printf("\n");
    return 0;
}
