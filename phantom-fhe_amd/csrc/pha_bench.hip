// pha_bench.hip -- measurement hooks (include/phantom_amd_bench.h): used by bench.py and tools/, not part of the drop-in boundary.
#include "../../include/phantom_amd_bench.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

using namespace pha;

namespace {

// Streaming kernels of the calibration (r04, tools/stream_calib.hip): ONE 16-byte word per lane, one trip per thread, a grid that
// covers the buffer -- the form that reaches the guide's 6.29 TB/s float4 copy on this part.  (The r01-r03 kernel kept four loads in
// flight per lane over a grid-stride loop and measured 4.8 TB/s on buffers beyond the MALL: more bytes in flight per lane run
// slower, profiles/r04_stream_calibration.txt.)  MODE 0 copy, 1 read-only, 2 write-only, 3 in-place read-modify-write; NT = nontemporal.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t words16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words16) return;
    u32x4 v = {1u, 2u, 3u, 4u};
    if (MODE != 2) v = NT ? __builtin_nontemporal_load((MODE == 3 ? (const u32x4 *)dst : src) + i) : (MODE == 3 ? (const u32x4 *)dst : src)[i];
    if (MODE == 1) {
        if (v.x + v.y + v.z + v.w == 0x12345678u && v.x == 0x9abcdef0u) dst[0] = v;   // keeps the load alive; practically never true
        return;
    }
    if (MODE == 3) v += 1u;
    if (NT) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
}

void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

}  // namespace

extern "C" {

int pha_repeat_forward_ntt_batched(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start, size_t batch,
                                   size_t poly_stride, int repeats, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (batch == 0 || batch > 65535) throw std::invalid_argument("batch out of range");
    NttExtra x;
    x.batch = (uint32_t)batch;
    x.poly_stride = poly_stride;
    for (int i = 0; i < repeats; i++)
        ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, x, as_stream(stream));
    PHA_API_END
}

int pha_time_forward_ntt(pha_context_t ctx, uint64_t *inout, size_t cms, int iters, void *stream, float *ms_out) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    hipStream_t s = as_stream(stream);
    hipEvent_t e0, e1;
    PHA_HIP(hipEventCreate(&e0));
    PHA_HIP(hipEventCreate(&e1));
    PHA_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++)
        ntt_forward(ctx->c, inout, inout, inout, plain_sel(0, cms), EPI_FWD_CANON, NttExtra{}, s);
    PHA_HIP(hipEventRecord(e1, s));
    PHA_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PHA_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / (float)iters;
    PHA_API_END
}

int pha_context_arena_count(pha_context_t ctx, size_t *count) {
    PHA_CTX_BEGIN(ctx)
    need(count);
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    *count = ctx->c.arenas.size() + ctx->c.outer_arenas.size();
    PHA_API_END
}

int pha_time_stream(uint64_t *dst, const uint64_t *src, size_t bytes, int mode, int nontemporal, int iters, void *stream,
                    double *bytes_per_s) {
    PHA_API_BEGIN
    need(dst); need(bytes_per_s);
    if (mode != 2 && mode != 3) need(src);
    if (bytes == 0 || bytes % 16 || iters < 1 || mode < 0 || mode > 3) throw std::invalid_argument("bytes must be a positive multiple of 16, iters >= 1, mode 0..3");
    hipStream_t s = as_stream(stream);
    const size_t words16 = bytes / 16;
    const unsigned blocks = (unsigned)((words16 + 255) / 256);
    u32x4 *d = reinterpret_cast<u32x4 *>(dst);
    const u32x4 *sp = reinterpret_cast<const u32x4 *>(src);
    auto launch = [&]() {
        switch (mode * 2 + (nontemporal ? 1 : 0)) {
            case 0: hipLaunchKernelGGL((stream_kernel<0, false>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 1: hipLaunchKernelGGL((stream_kernel<0, true>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 2: hipLaunchKernelGGL((stream_kernel<1, false>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 3: hipLaunchKernelGGL((stream_kernel<1, true>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 4: hipLaunchKernelGGL((stream_kernel<2, false>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 5: hipLaunchKernelGGL((stream_kernel<2, true>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            case 6: hipLaunchKernelGGL((stream_kernel<3, false>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
            default: hipLaunchKernelGGL((stream_kernel<3, true>), dim3(blocks), dim3(256), 0, s, d, sp, words16); break;
        }
        check_launch();
    };
    for (int i = 0; i < 3; i++) launch();
    hipEvent_t e0, e1;
    PHA_HIP(hipEventCreate(&e0));
    PHA_HIP(hipEventCreate(&e1));
    PHA_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) launch();
    PHA_HIP(hipEventRecord(e1, s));
    PHA_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PHA_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *bytes_per_s = ((mode == 0 || mode == 3) ? 2.0 : 1.0) * (double)bytes * iters / ((double)ms * 1e-3);
    PHA_API_END
}

int pha_time_stream_copy(uint64_t *dst, const uint64_t *src, size_t bytes, int iters, void *stream, double *bytes_per_s) {
    return pha_time_stream(dst, src, bytes, 0, 1, iters, stream, bytes_per_s);
}

}  // extern "C"
