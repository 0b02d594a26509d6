// pattern_bench.hip -- r04: the memory floor of the NTT passes' access patterns on an HBM-resident batch (16 polynomials x 45 limbs x
// 65536 coefficients = 360 MiB, in-place read-modify-write, no butterflies, no LDS), against the linear stream of tools/stream_calib.hip.
//   lin16   : 256-thread workgroups, one 16-byte word per lane (the calibration's best form)
//   lin8x8  : one wavefront per 512-coefficient tile, 8 loads of 8 B per lane (512 B per instruction), same pattern back
//   cont    : the contiguous pass's exact pattern (loads: two 256-byte half-rows per instruction; stores: 32 B per lane as 2 x 16 B)
//   cont16  : the same tile with 16-byte loads (4 per lane)
//   strided : the strided pass's exact pattern (512 threads, 256 rows x 16 columns: 128-byte runs at a 2 KiB stride)
//   strided32: 128 rows x 32 columns (256-byte runs at a 4 KiB stride)
// Block orders for the one-wavefront tiles: linear (tile, limb, polynomial) and the product's polynomial-fastest XCD-grouped order.
// DELAY: cycles every wavefront waits between its loads and its stores (s_sleep loop), standing in for the butterflies.
// Build: hipcc -O3 --offload-arch=gfx950 tools/pattern_bench.hip -o tools/pattern_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
typedef unsigned long long u64v2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int N = 65536, LIMBS = 45, POLYS = 16;

template <bool NT> __device__ __forceinline__ u64 ld(const u64 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u64 *p, u64 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ u64v2 ld2(const u64 *p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const u64v2 *>(p)) : *reinterpret_cast<const u64v2 *>(p);
}
template <bool NT> __device__ __forceinline__ void st2(u64 *p, u64v2 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u64v2 *>(p)); else *reinterpret_cast<u64v2 *>(p) = v;
}
__device__ __forceinline__ void delay(int cycles) {
    if (cycles <= 0) return;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}

// one-wavefront tile: 512 coefficients = rows (2 t, 2 t + 1) of the 256 x 256 view of a limb.  ORDER 0: blockIdx.x = tile (128 per
// limb), y = limb, z = polynomial; ORDER 1: the product's 1-D polynomial-fastest order
template <int MODE, bool NT, int ORDER>
__global__ __launch_bounds__(64) void wave_tile_kernel(u64 *buf, int cycles) {
    unsigned tile = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (ORDER == 1) {
        const unsigned b = blockIdx.x, q = b >> 3;
        z = q % POLYS;
        const unsigned group = (q / POLYS) * 8 + (b & 7u);
        tile = group % 128;
        y = group / 128;
    }
    u64 *p = buf + ((size_t)z * LIMBS + y) * N + (size_t)tile * 512;
    const int lane = threadIdx.x;
    u64 v[8];
    if (MODE == 0) {   // lin8x8
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = ld<NT>(p + lane + 64 * k);
        delay(cycles);
#pragma unroll
        for (int k = 0; k < 8; k++) st<NT>(p + lane + 64 * k, v[k] + 1);
    } else if (MODE == 1) {   // cont: the pass's own maps
        const int row = lane >> 5, lo = lane & 31;
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = ld<NT>(p + row * 256 + lo + 32 * k);
        delay(cycles);
#pragma unroll
        for (int gi = 0; gi < 2; gi++) {
            st2<NT>(p + gi * 256 + 4 * lane, u64v2{v[4 * gi] + 1, v[4 * gi + 1] + 1});
            st2<NT>(p + gi * 256 + 4 * lane + 2, u64v2{v[4 * gi + 2] + 1, v[4 * gi + 3] + 1});
        }
    } else {   // cont16: 16-byte loads and stores, 1 KiB per instruction
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64v2 t = ld2<NT>(p + 2 * lane + 128 * k);
            v[2 * k] = t.x;
            v[2 * k + 1] = t.y;
        }
        delay(cycles);
#pragma unroll
        for (int k = 0; k < 4; k++) st2<NT>(p + 2 * lane + 128 * k, u64v2{v[2 * k] + 1, v[2 * k + 1] + 1});
    }
}

// strided tile: 4096 coefficients = ROWS rows x V columns, 512 threads x 8
template <int LOGV, bool NT>
__global__ __launch_bounds__(512) void strided_tile_kernel(u64 *buf, int cycles) {
    constexpr int V = 1 << LOGV, ROWS = 4096 / V, T2 = N / ROWS;   // row stride in elements
    u64 *p = buf + ((size_t)blockIdx.z * LIMBS + blockIdx.y) * N + (size_t)blockIdx.x * V;
    const int t = threadIdx.x, c = t & (V - 1), r0 = t >> LOGV;
    constexpr int RSTEP = 512 / V;
    u64 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ld<NT>(p + (size_t)(r0 + RSTEP * k) * T2 + c);
    delay(cycles);
    // last round's layout: 4 consecutive rows per group, two groups per thread
#pragma unroll
    for (int gi = 0; gi < 2; gi++)
#pragma unroll
        for (int k = 0; k < 4; k++) st<NT>(p + (size_t)(4 * (r0 + RSTEP * gi) + k) % ROWS * T2 + c, v[4 * gi + k] + 1);
}

template <bool NT>
__global__ __launch_bounds__(256) void lin16_kernel(u64 *buf, int cycles) {
    u64 *p = buf + ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const u64v2 t = ld2<NT>(p);
    delay(cycles);
    st2<NT>(p, u64v2{t.x + 1, t.y + 1});
}

template <class F>
static void timeit(const char *name, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(a));
    const int reps = 30;
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 2.0 * POLYS * LIMBS * N * 8;
    printf("%-44s %7.1f us  %6.0f GB/s (r+w)\n", name, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}

int main() {
    u64 *buf;
    const size_t words = (size_t)POLYS * LIMBS * N;
    CK(hipMalloc(&buf, words * 8));
    CK(hipMemset(buf, 0, words * 8));
    for (int cycles : {0, 2000, 6000}) {
        printf("---- delay between loads and stores: %d cycles\n", cycles);
        const dim3 g3(128, LIMBS, POLYS), g1(128 * LIMBS * POLYS);
        timeit("lin16", [&] { lin16_kernel<false><<<(unsigned)(words / 512), 256>>>(buf, cycles); });
        timeit("lin16 nt", [&] { lin16_kernel<true><<<(unsigned)(words / 512), 256>>>(buf, cycles); });
        timeit("lin8x8 linear order", [&] { wave_tile_kernel<0, false, 0><<<g3, 64>>>(buf, cycles); });
        timeit("lin8x8 linear order nt", [&] { wave_tile_kernel<0, true, 0><<<g3, 64>>>(buf, cycles); });
        timeit("cont linear order", [&] { wave_tile_kernel<1, false, 0><<<g3, 64>>>(buf, cycles); });
        timeit("cont linear order nt", [&] { wave_tile_kernel<1, true, 0><<<g3, 64>>>(buf, cycles); });
        timeit("cont polynomial-fastest order", [&] { wave_tile_kernel<1, false, 1><<<g1, 64>>>(buf, cycles); });
        timeit("cont polynomial-fastest order nt", [&] { wave_tile_kernel<1, true, 1><<<g1, 64>>>(buf, cycles); });
        timeit("cont16 linear order", [&] { wave_tile_kernel<2, false, 0><<<g3, 64>>>(buf, cycles); });
        timeit("cont16 linear order nt", [&] { wave_tile_kernel<2, true, 0><<<g3, 64>>>(buf, cycles); });
        timeit("strided 256 x 16 (128 B runs)", [&] { strided_tile_kernel<4, false><<<dim3(16, LIMBS, POLYS), 512>>>(buf, cycles); });
        timeit("strided 256 x 16 nt", [&] { strided_tile_kernel<4, true><<<dim3(16, LIMBS, POLYS), 512>>>(buf, cycles); });
        timeit("strided 128 x 32 (256 B runs)", [&] { strided_tile_kernel<5, false><<<dim3(16, LIMBS, POLYS), 512>>>(buf, cycles); });
        timeit("strided 128 x 32 nt", [&] { strided_tile_kernel<5, true><<<dim3(16, LIMBS, POLYS), 512>>>(buf, cycles); });
        timeit("strided 64 x 64 (512 B runs)", [&] { strided_tile_kernel<6, false><<<dim3(16, LIMBS, POLYS), 512>>>(buf, cycles); });
    }
    return 0;
}
