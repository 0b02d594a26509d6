// stream_calib.hip -- independent calibration of the streaming ceiling of the box.
//
// Measures, for buffers inside the 256 MiB Infinity Cache (MALL) and far outside it:
//   read-only   (16 B per lane, summed into a register; one store per workgroup)
//   write-only  (16 B per lane)
//   copy 1R:1W  in several forms, incl. the float4 copy of MI355X_MICROARCH.md (6.29 TB/s there)
//   in-place    read-modify-write of the same addresses (what an in-place NTT pass does)
// Forms (template parameters): ILP = loads in flight per lane (1 / 4 / 8), NT = nontemporal policy on loads and stores,
// grid = "cover" (one trip per thread: blocks = bytes / (256 * 16 * ILP)) or "persist" (#CU x k workgroups, grid-stride).
// Rates are (bytes read + bytes written) / time, averaged over `iters` launches after 3 warm-ups; hipMemcpyDtoD beside them.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_calib.hip -o tools/stream_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4 *p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st(f32x4 *p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// OP: 0 copy, 1 read-only, 2 write-only, 3 in-place read-modify-write
template <int OP, int ILP, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(f32x4 *__restrict__ dst, const f32x4 *__restrict__ src, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += ILP * stride) {
        f32x4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            const size_t j = i + k * stride;
            if (OP != 2) v[k] = j < words ? ld<NT>((OP == 3 ? (const f32x4 *)dst : src) + j) : acc;
            else v[k] = f32x4{1.f, 2.f, 3.f, (float)k};
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            const size_t j = i + k * stride;
            if (OP == 1) acc += v[k];
            else if (j < words) st<NT>(dst + j, OP == 3 ? v[k] + 1.0f : v[k]);
        }
    }
    if (OP == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[blockIdx.x] = acc;   // keeps the loads alive; never true
}

struct Result {
    std::string name;
    double gbps;
};

template <int OP, int ILP, bool NT>
static double run(f32x4 *dst, const f32x4 *src, size_t bytes, int blocks_per_cu, int iters) {
    const size_t words = bytes / 16;
    size_t blocks = blocks_per_cu > 0 ? (size_t)256 * blocks_per_cu : (words + 256 * ILP - 1) / (256 * ILP);
    if (blocks > (words + 255) / 256) blocks = (words + 255) / 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) stream_kernel<OP, ILP, NT><<<dim3((unsigned)blocks), 256>>>(dst, src, words);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) stream_kernel<OP, ILP, NT><<<dim3((unsigned)blocks), 256>>>(dst, src, words);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    const double moved = (OP == 0 || OP == 3) ? 2.0 * bytes : (double)bytes;
    return moved * iters / (ms * 1e-3) / 1e9;
}

template <int OP>
static void sweep(const char *what, f32x4 *dst, const f32x4 *src, size_t bytes, int iters) {
    printf("  %-12s", what);
    // cover grid (one trip per thread), ILP 1 / 4 / 8, default and nt; persistent 8 / 16 / 32 workgroups per CU with ILP 4
    printf(" cover: ilp1 %6.0f  ilp4 %6.0f  ilp8 %6.0f | nt ilp1 %6.0f  ilp4 %6.0f  ilp8 %6.0f", run<OP, 1, false>(dst, src, bytes, 0, iters),
           run<OP, 4, false>(dst, src, bytes, 0, iters), run<OP, 8, false>(dst, src, bytes, 0, iters), run<OP, 1, true>(dst, src, bytes, 0, iters),
           run<OP, 4, true>(dst, src, bytes, 0, iters), run<OP, 8, true>(dst, src, bytes, 0, iters));
    printf(" | persist ilp4: x8 %6.0f  x16 %6.0f  x32 %6.0f | nt x8 %6.0f  x16 %6.0f  x32 %6.0f | persist ilp8 x4 %6.0f nt %6.0f\n",
           run<OP, 4, false>(dst, src, bytes, 8, iters), run<OP, 4, false>(dst, src, bytes, 16, iters), run<OP, 4, false>(dst, src, bytes, 32, iters),
           run<OP, 4, true>(dst, src, bytes, 8, iters), run<OP, 4, true>(dst, src, bytes, 16, iters), run<OP, 4, true>(dst, src, bytes, 32, iters),
           run<OP, 8, false>(dst, src, bytes, 4, iters), run<OP, 8, true>(dst, src, bytes, 4, iters));
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, memory clock %d kHz, bus %d bits; GB/s = (read + written bytes) / time\n", prop.name, prop.multiProcessorCount,
           prop.memoryClockRate, prop.memoryBusWidth);
    const size_t max_bytes = (size_t)2048 << 20;
    f32x4 *a, *b;
    CK(hipMalloc(&a, max_bytes));
    CK(hipMalloc(&b, max_bytes));
    CK(hipMemset(a, 1, max_bytes));
    CK(hipMemset(b, 2, max_bytes));
    const size_t sizes_mib[] = {32, 90, 180, 360, 512, 1024, 2048};
    for (size_t mib : sizes_mib) {
        const size_t bytes = mib << 20;
        const int iters = mib <= 180 ? 200 : mib <= 512 ? 60 : 20;
        printf("buffer %zu MiB per operand\n", mib);
        sweep<0>("copy a->b", b, a, bytes, iters);
        sweep<1>("read-only", b, a, bytes, iters);
        sweep<2>("write-only", b, a, bytes, iters);
        sweep<3>("in-place rmw", b, a, bytes, iters);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  hipMemcpyDtoD %6.0f\n", 2.0 * bytes * iters / (ms * 1e-3) / 1e9);
    }
    // the two-pass pattern in miniature: pass A streams a chunk in place, pass B streams the same chunk in place again; chunk sizes
    // inside and outside the MALL over a 2 GiB buffer (every chunk is touched once per sweep: HBM-resident at its first read)
    printf("two in-place passes per chunk over a 1440 MiB buffer (GB/s counts 2 reads + 2 writes per byte; algorithmic = half)\n");
    const size_t total = (size_t)1440 << 20;
    for (size_t chunk_mib : {1440, 720, 360, 180, 120, 90, 60, 45}) {
        const size_t chunk = chunk_mib << 20;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int reps = 4;
        for (int w = 0; w < 2; w++) {
            if (w == 1) CK(hipEventRecord(e0));
            for (int r = 0; r < (w ? reps : 1); r++)
                for (size_t off = 0; off < total; off += chunk) {
                    f32x4 *p = (f32x4 *)((char *)a + off);
                    const size_t words = chunk / 16;
                    const unsigned blocks = (unsigned)((words + 256 * 4 - 1) / (256 * 4));
                    stream_kernel<3, 4, false><<<blocks, 256>>>(p, p, words);
                    stream_kernel<3, 4, false><<<blocks, 256>>>(p, p, words);
                }
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  chunk %5zu MiB: %6.0f GB/s through the kernels, %6.0f GB/s algorithmic (1R + 1W per byte)\n", chunk_mib,
               4.0 * total * reps / (ms * 1e-3) / 1e9, 2.0 * total * reps / (ms * 1e-3) / 1e9);
    }
    return 0;
}
