// pha_bench.hip -- measurement hooks (include/phantom_amd_bench.h): used by bench.py and tools/, not part of the drop-in boundary.
#include "../../include/phantom_amd_bench.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

using namespace pha;

namespace {

// 16 bytes per lane, grid-stride, four loads in flight per lane: the plain streaming pattern of the guide's 6.29 TB/s float4 copy
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t words16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < words16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                    c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < words16; i += stride) dst[i] = src[i];
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

int pha_time_stream_copy(uint64_t *dst, const uint64_t *src, size_t bytes, int iters, void *stream, double *bytes_per_s) {
    PHA_API_BEGIN
    need(dst); need(src); need(bytes_per_s);
    if (bytes == 0 || bytes % 16 || iters < 1) throw std::invalid_argument("bytes must be a positive multiple of 16, iters >= 1");
    hipStream_t s = as_stream(stream);
    const size_t words16 = bytes / 16;
    const unsigned blocks = (unsigned)std::min<size_t>((words16 + 255) / 256, 256 * 32);   // 32 workgroups per CU, grid-stride
    auto launch = [&]() {
        hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<u32x4 *>(dst),
                           reinterpret_cast<const u32x4 *>(src), words16);
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
    *bytes_per_s = 2.0 * (double)bytes * iters / ((double)ms * 1e-3);
    PHA_API_END
}

}  // extern "C"
