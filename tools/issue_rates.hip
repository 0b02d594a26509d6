// issue_rates.hip -- r06: issue cost of the vector instructions the conversion / butterfly kernels are made of, per wave64
// instruction and SIMD, in SHADER-CLOCK cycles (s_memtime deltas inside the kernel; s_memrealtime beside it gives the clock the
// part actually runs at under this load).  Every instruction is an `asm volatile`, eight independent chains per thread, so the
// compiler neither removes nor reorders them.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rates tools/issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint64_t u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
enum { I_MAD64, I_MULLO, I_LSHLADD64, I_LSHR64, I_AND32, I_ADD32, I_CNDMASK, I_CMP64, I_ADDCO, I_FMA64, I_MUL64F, I_ADD64F, I_RNDNE64, I_MOV32,
       I_DSREAD_BCAST, I_DSREAD_LANE, I_MAD_THEN_SHIFT, I_MAD_THEN_ADD, I_FMA32, I_MAD_DEP, I_FMA64_DEP, I_COUNT };
static const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_lshl_add_u64", "v_lshrrev_b64", "v_and_b32", "v_add_u32", "v_cndmask_b32", "v_cmp_lt_u64",
                              "v_add_co_u32 + v_addc_co_u32 (pair)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_mov_b32",
                              "ds_read_b64 (one address)", "ds_read_b64 (per lane)", "v_mad_u64_u32 -> v_lshrrev_b64 of it (pair)",
                              "v_mad_u64_u32 -> v_lshl_add_u64 of it (pair)", "v_fma_f32", "v_mad_u64_u32 chain (dependent)", "v_fma_f64 chain (dependent)"};
constexpr int CH = 8, INNER = 32;
template <int OP>
__global__ __launch_bounds__(256) void k(u64 *out, u64 *clk, int iters, u64 seed) {
    __shared__ u64 lds[512];
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 512; i += 256) lds[i] = seed + i;
    __syncthreads();
    u64 a[CH], b[CH];
    u32 x[CH], y[CH];
    double d[CH], e[CH];
    float f[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        a[c] = seed * (tid + 1) + c; b[c] = a[c] ^ (seed >> 3); x[c] = (u32)a[c]; y[c] = (u32)b[c] | 1; d[c] = (double)(x[c] & 1023); e[c] = 1.0000001; f[c] = (float)(x[c] & 255);
    }
    const u32 laddr = (OP == I_DSREAD_LANE ? (threadIdx.x & 63) * 8 : 0);
    const u64 t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < INNER; j++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                u64 cy;
                if (OP == I_MAD64) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(a[c]), "=s"(cy) : "v"(x[c]), "v"(y[c]), "v"(b[c]));
                else if (OP == I_MAD_DEP) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a[c]), "=s"(cy) : "v"(x[c]), "v"(y[c]));
                else if (OP == I_MULLO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"((u32)b[c]));
                else if (OP == I_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[c]) : "v"(b[c]), "v"(a[(c + 1) % CH]));
                else if (OP == I_LSHR64) asm volatile("v_lshrrev_b64 %0, 30, %1" : "=v"(a[c]) : "v"(b[c]));
                else if (OP == I_AND32) asm volatile("v_and_b32 %0, 0x3fffffff, %1" : "=v"(x[c]) : "v"(y[c]));
                else if (OP == I_ADD32) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"((u32)b[c]));
                else if (OP == I_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[c]) : "v"(y[c]), "v"((u32)b[c]) : "vcc");
                else if (OP == I_CMP64) asm volatile("v_cmp_lt_u64 vcc, %0, %1" ::"v"(a[c]), "v"(b[c]) : "vcc");
                else if (OP == I_ADDCO) asm volatile("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc" : "=&v"(x[c]), "=v"(y[c]) : "v"((u32)a[c]), "v"((u32)b[c]), "v"((u32)(a[c] >> 32)), "v"((u32)(b[c] >> 32)) : "vcc");
                else if (OP == I_FMA64) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[c]) : "v"(e[c]), "v"(e[(c + 1) % CH]), "v"(e[(c + 2) % CH]));
                else if (OP == I_FMA64_DEP) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(e[c]), "v"(e[(c + 2) % CH]));
                else if (OP == I_MUL64F) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[c]) : "v"(e[c]), "v"(e[(c + 1) % CH]));
                else if (OP == I_ADD64F) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[c]) : "v"(e[c]), "v"(e[(c + 1) % CH]));
                else if (OP == I_RNDNE64) asm volatile("v_rndne_f64 %0, %1" : "=v"(d[c]) : "v"(e[c]));
                else if (OP == I_MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(y[c]));
                else if (OP == I_FMA32) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[c]) : "v"(f[(c + 1) % CH]), "v"(f[(c + 2) % CH]), "v"(f[(c + 3) % CH]));
                else if (OP == I_DSREAD_BCAST || OP == I_DSREAD_LANE) asm volatile("ds_read_b64 %0, %1" : "=v"(a[c]) : "v"(laddr + 8 * c));
                else if (OP == I_MAD_THEN_SHIFT) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4\n\ts_nop 0\n\tv_lshrrev_b64 %0, 30, %0" : "=&v"(a[c]), "=s"(cy) : "v"(x[c]), "v"(y[c]), "v"(b[c]));
                else if (OP == I_MAD_THEN_ADD) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4\n\ts_nop 0\n\tv_lshl_add_u64 %0, %0, 0, %4" : "=&v"(a[c]), "=s"(cy) : "v"(x[c]), "v"(y[c]), "v"(b[c]));
            }
        }
        if (OP == I_DSREAD_BCAST || OP == I_DSREAD_LANE) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const u64 t1 = clock64(), w1 = wall_clock64();
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc += a[c] + b[c] + x[c] + y[c] + (u64)d[c] + (u64)f[c];
    out[tid] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int OP>
static int run(u64 *out, u64 *clk, int blocks_per_cu, int cus) {
    const int blocks = blocks_per_cu * cus, iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 2, 0x123456789abcdefull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 0x123456789abcdefull);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<u64> h(2 * blocks);
    CK(hipMemcpy(h.data(), clk, sizeof(u64) * 2 * blocks, hipMemcpyDeviceToHost));
    double sc = 0, wc = 0;
    for (int b = 0; b < blocks; b++) { sc += (double)h[2 * b]; wc += (double)h[2 * b + 1]; }
    sc /= blocks; wc /= blocks;                       // per workgroup: shader-clock cycles, 100 MHz ticks
    const double ghz = sc / (wc * 10.0);              // shader clock while this kernel ran
    const int per = (OP == I_ADDCO || OP == I_MAD_THEN_SHIFT || OP == I_MAD_THEN_ADD) ? 2 : 1;
    // wave-instructions one SIMD issued while a workgroup was resident: blocks_per_cu workgroups x 4 waves / 4 SIMDs = blocks_per_cu waves per SIMD
    const double wave_instr_per_simd = (double)blocks_per_cu * iters * INNER * CH * per;
    printf("%-46s %2d waves/SIMD  %7.3f ms  clock %.2f GHz  %6.2f cycles per wave-instruction and SIMD\n", names[OP], blocks_per_cu, ms, ghz, sc / wave_instr_per_simd);
    return 0;
}
template <int OP>
static int both(u64 *out, u64 *clk, int cus) { return run<OP>(out, clk, 1, cus) || run<OP>(out, clk, 2, cus) || run<OP>(out, clk, 4, cus); }
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  nominal clock %.2f GHz\n", p.name, cus, p.clockRate / 1e6);
    u64 *out, *clk; CK(hipMalloc(&out, sizeof(u64) * cus * 4 * 256)); CK(hipMalloc(&clk, sizeof(u64) * 2 * cus * 4));
    int rc = 0;
    rc |= both<I_MAD64>(out, clk, cus); rc |= both<I_MAD_DEP>(out, clk, cus); rc |= both<I_MULLO>(out, clk, cus); rc |= both<I_LSHLADD64>(out, clk, cus); rc |= both<I_LSHR64>(out, clk, cus);
    rc |= both<I_AND32>(out, clk, cus); rc |= both<I_ADD32>(out, clk, cus); rc |= both<I_CNDMASK>(out, clk, cus); rc |= both<I_CMP64>(out, clk, cus);
    rc |= both<I_ADDCO>(out, clk, cus); rc |= both<I_FMA64>(out, clk, cus); rc |= both<I_FMA64_DEP>(out, clk, cus); rc |= both<I_MUL64F>(out, clk, cus); rc |= both<I_ADD64F>(out, clk, cus);
    rc |= both<I_RNDNE64>(out, clk, cus); rc |= both<I_MOV32>(out, clk, cus); rc |= both<I_FMA32>(out, clk, cus); rc |= both<I_DSREAD_BCAST>(out, clk, cus); rc |= both<I_DSREAD_LANE>(out, clk, cus);
    rc |= both<I_MAD_THEN_SHIFT>(out, clk, cus); rc |= both<I_MAD_THEN_ADD>(out, clk, cus);
    return rc;
}
