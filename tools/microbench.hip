// microbench.hip -- gfx950 instruction-rate and bandwidth probes that size the NTT design
// (is the 64-bit modular butterfly ALU-bound before it is HBM-bound?).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint64_t u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

enum { OP_FMA32, OP_ADD32, OP_MULLO, OP_MULHI, OP_MAD64, OP_ADD64, OP_FMA64, OP_MUL64LO, OP_MUL64HI, OP_SHOUP, OP_BFLY, OP_BFLY_A, OP_BFLY_B, OP_BFLY_C, OP_LSHLADD, OP_FP64BF };
static const char *names[] = {"v_fma_f32", "v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "add64(v_lshl_add_u64)",
                              "v_fma_f64", "mul64 lo (a*b)", "mul64 hi (__umul64hi)", "shoup_lazy", "ct butterfly", "bfly A: approx mulhi (asm mad)", "bfly B: A + negq mad chain", "bfly C: B + sign-mask csub", "v_lshl_add_u64 (asm)", "fp64 butterfly (11 ops)"};
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) { u64 d, cy; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ u64 mad64s(u32 a, u32 b, u64 c) { u64 d, cy; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c)); return d; }
__device__ __forceinline__ u64 mad64z(u32 a, u32 b) { u64 d, cy; asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u64 mad64one(u32 a, u64 c) { u64 d, cy; asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cy) : "v"(a), "v"(c)); return d; }
__device__ __forceinline__ u64 mulhi_approx(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p01 = mad64z(a0, b1), p10 = mad64z(a1, b0);
    return mad64one((u32)(p10 >> 32), mad64(a1, b1, p01 >> 32));
}
__device__ __forceinline__ u64 shoup4_negq(u64 Y, u64 w, u64 ws, u64 nq) {
    u32 y0 = (u32)Y, y1 = (u32)(Y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    u64 Q = mulhi_approx(Y, ws);
    u32 q0 = (u32)Q, q1 = (u32)(Q >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    u64 T = mad64s(q0, n0, mad64z(y0, w0));
    u32 hi = (u32)(T >> 32) + y0 * w1 + y1 * w0 + q0 * n1 + q1 * n0;
    return ((u64)hi << 32) | (u32)T;
}
constexpr int CH = 8;      // independent chains per thread
constexpr int INNER = 64;  // ops per chain per outer iteration

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(u64 *out, u64 seed, int iters) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[CH], b[CH];
    float fa[CH]; double da[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { a[c] = seed * (tid + 1) + c * 0x9e3779b97f4a7c15ull; b[c] = a[c] ^ (seed >> 3); fa[c] = (float)(a[c] & 1023); da[c] = (double)(a[c] & 1023); }
    const u64 q = (seed | 1) & ((1ull << 60) - 1), q2 = q * 2;
    const u64 w = seed * 3 + 1, ws = seed * 7 + 5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < INNER; j++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == OP_FMA32) fa[c] = __builtin_fmaf(fa[c], 1.0001f, 0.5f);
                else if (OP == OP_ADD32) { u32 x = (u32)a[c]; x = x + (u32)b[c]; a[c] = x; }
                else if (OP == OP_MULLO) { u32 x = (u32)a[c]; x = x * (u32)b[c] + 1; a[c] = x; }
                else if (OP == OP_MULHI) { u32 x = (u32)a[c]; x = __umulhi(x, (u32)b[c]) | 0x10001u; a[c] = x; }
                else if (OP == OP_MAD64) { a[c] = (u64)(u32)a[c] * (u64)(u32)b[c] + a[c]; }
                else if (OP == OP_ADD64) { a[c] = a[c] + b[c]; }
                else if (OP == OP_FMA64) da[c] = __builtin_fma(da[c], 1.0000001, 0.5);
                else if (OP == OP_MUL64LO) a[c] = a[c] * b[c] + 1;
                else if (OP == OP_MUL64HI) a[c] = __umul64hi(a[c], b[c]) | 0x100000001ull;
                else if (OP == OP_SHOUP) a[c] = a[c] * w - __umul64hi(a[c], ws) * q;
                else if (OP == OP_BFLY_A) {
                    u64 X = a[c], Y = b[c]; const u64 q4 = q2 * 2;
                    u64 x = X >= q4 ? X - q4 : X;
                    u64 t = Y * w - mulhi_approx(Y, ws) * q;
                    a[c] = x + t; b[c] = x + q4 - t;
                } else if (OP == OP_BFLY_B) {
                    u64 X = a[c], Y = b[c]; const u64 q4 = q2 * 2;
                    u64 x = X >= q4 ? X - q4 : X;
                    u64 t = shoup4_negq(Y, w, ws, 0 - q);
                    a[c] = x + t; b[c] = x + q4 - t;
                } else if (OP == OP_BFLY_C) {
                    u64 X = a[c], Y = b[c]; const u64 q4 = q2 * 2;
                    u64 d = X + (0 - q4);
                    u32 mask = (u32)((int32_t)(d >> 32) >> 31);
                    u64 x = d + ((((u64)((u32)(q4 >> 32) & mask)) << 32) | ((u32)q4 & mask));
                    u64 t = shoup4_negq(Y, w, ws, 0 - q);
                    a[c] = x + t; b[c] = x + q4 - t;
                } else if (OP == OP_LSHLADD) {
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(b[c]));
                } else if (OP == OP_FP64BF) {
                    double X = da[c], Y = __longlong_as_double(b[c] & 0x3fffffffffffffffull | 0x3ff0000000000000ull);
                    const double W = 1234567.0, Wi = 1e-9, qd = 1125899906842597.0, qi = 1.0 / 1125899906842597.0;
                    double h = Y * W, l = __builtin_fma(Y, W, -h);
                    double cq = __builtin_rint(Y * Wi);
                    double r = __builtin_fma(-cq, qd, h) + l;
                    double c2 = __builtin_rint(r * qi);
                    double T = __builtin_fma(-c2, qd, r);
                    da[c] = X + T; b[c] = __double_as_longlong(X - T);
                }
                else if (OP == OP_BFLY) {
                    u64 X = a[c], Y = b[c];
                    u64 x = X >= q2 ? X - q2 : X;
                    u64 t = Y * w - __umul64hi(Y, ws) * q;
                    a[c] = x + t; b[c] = x + q2 - t;
                }
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc += a[c] + b[c] + (u64)fa[c] + (u64)da[c];
    out[tid] = acc;
}

template <int OP>
static int run_rate(u64 *out, int blocks, double clk_ghz) {
    const int iters = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 0x123456789abcdefull, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 0x123456789abcdefull, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 256 * iters * INNER * CH;            // lane-ops
    const double wave_ops_per_simd = ops / 64.0 / (256.0 * 4);                // wave-instructions per SIMD
    const double ns_per = ms * 1e6 / wave_ops_per_simd;
    printf("%-26s %8.3f ms  %8.2f Tlane-op/s  %6.2f ns/wave-op/SIMD = %6.2f cyc @%.2fGHz\n", names[OP], ms, ops / ms / 1e9,
           ns_per, ns_per * clk_ghz, clk_ghz);
    return 0;
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ __launch_bounds__(256) void rw_kernel(uint4 *__restrict__ buf, size_t n16) {  // in-place read+write
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = buf[i]; v.x += 1; buf[i] = v;
    }
}

// Memory floor of the NTT's access patterns with the butterflies removed: one workgroup of 512 threads per
// 4096-coefficient tile, 8 coefficients per thread, in-place read-modify-write of a [limbs][65536] buffer.
// MODE 0: strided pass (tile = 16 adjacent columns x 256 rows of the 256x256 view, 8 B per lane)
// MODE 1: contiguous pass, 8 B per lane (thread t touches t + 512 j)
// MODE 2: contiguous pass, 16 B per lane (thread t touches the pairs 2t + 1024 j)
template <int MODE>
__global__ __launch_bounds__(512) void tile_rw_kernel(u64 *buf) {
    u64 *limb = buf + (size_t)blockIdx.y * 65536;
    const uint32_t t = threadIdx.x, tile = blockIdx.x;
    u64 v[8];
    if (MODE == 0) {
        const uint32_t c = tile * 16 + (t & 15), r0 = t >> 4;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = limb[(r0 + 32 * j) * 256 + c];
#pragma unroll
        for (int j = 0; j < 8; j++) limb[(r0 + 32 * j) * 256 + c] = v[j] + 1;
    } else if (MODE == 1) {
        u64 *p = limb + tile * 4096 + t;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[512 * j];
#pragma unroll
        for (int j = 0; j < 8; j++) p[512 * j] = v[j] + 1;
    } else {
        uint4 *p = reinterpret_cast<uint4 *>(limb + tile * 4096) + t;
        uint4 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = p[512 * j];
#pragma unroll
        for (int j = 0; j < 4; j++) { w[j].x += 1; p[512 * j] = w[j]; }
    }
}
template <int MODE>
static int run_tile(u64 *buf, int limbs, const char *name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 50;
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(tile_rw_kernel<MODE>, dim3(16, limbs), dim3(512), 0, 0, buf);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(tile_rw_kernel<MODE>, dim3(16, limbs), dim3(512), 0, 0, buf);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * limbs * 65536 * 8;
    printf("tile pattern %-34s limbs %3d: %6.2f us/launch  %5.2f TB/s (r+w)\n", name, limbs, ms * 1000 / reps, bytes * reps / ms / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    if (getenv("MB_PATTERN")) {
        u64 *buf; CK(hipMalloc(&buf, (size_t)180 * 65536 * 8)); CK(hipMemset(buf, 1, (size_t)180 * 65536 * 8));
        for (int limbs : {45, 180}) {
            run_tile<0>(buf, limbs, "strided, 8 B/lane");
            run_tile<1>(buf, limbs, "contiguous, 8 B/lane");
            run_tile<2>(buf, limbs, "contiguous, 16 B/lane");
        }
        return 0;
    }
    const double clk = p.clockRate / 1e6;
    printf("device %s  CUs %d  clock %.2f GHz  L2 %d KiB\n", p.name, p.multiProcessorCount, clk, p.l2CacheSize / 1024);
    const int blocks = p.multiProcessorCount * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8/SIMD
    u64 *out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    run_rate<OP_FMA32>(out, blocks, clk); run_rate<OP_ADD32>(out, blocks, clk); run_rate<OP_MULLO>(out, blocks, clk);
    run_rate<OP_MULHI>(out, blocks, clk); run_rate<OP_MAD64>(out, blocks, clk); run_rate<OP_ADD64>(out, blocks, clk);
    run_rate<OP_FMA64>(out, blocks, clk); run_rate<OP_MUL64LO>(out, blocks, clk); run_rate<OP_MUL64HI>(out, blocks, clk);
    run_rate<OP_SHOUP>(out, blocks, clk); run_rate<OP_BFLY>(out, blocks, clk);
    run_rate<OP_BFLY_A>(out, blocks, clk); run_rate<OP_BFLY_B>(out, blocks, clk); run_rate<OP_BFLY_C>(out, blocks, clk); run_rate<OP_LSHLADD>(out, blocks, clk); run_rate<OP_FP64BF>(out, blocks, clk);

    if (getenv("MB_NO_BW")) return 0;
    const size_t sizes_mib[] = {8, 23, 45, 90, 180, 512, 2048};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mib : sizes_mib) {
        const size_t bytes = mib << 20, n16 = bytes / 16;
        uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        const int grid = p.multiProcessorCount * 16, reps = 20;
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n16); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double copy_tbs = 2.0 * bytes * reps / ms / 1e9;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rw_kernel, dim3(grid), dim3(256), 0, 0, a, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double rw_tbs = 2.0 * bytes * reps / ms / 1e9;
        printf("buffer %5zu MiB: copy a->b %6.2f TB/s (r+w, %.1f us/launch)   in-place rw %6.2f TB/s (%.1f us/launch)\n", mib, copy_tbs,
               2.0 * bytes / copy_tbs / 1e6, rw_tbs, ms * 1000 / reps);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
