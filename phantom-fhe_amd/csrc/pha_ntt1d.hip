// Single-workgroup negacyclic NTT for small degrees (N <= 2048): the reference's fnwt_1d / fnwt_1d_opt / inwt_1d /
// inwt_1d_opt (include/ntt.cuh:157-171, src/ntt/ntt_1d.cu:17-292), which exist for its NTT test and benchmark
// (test/ntt_test.cu:9-69, benchmark/ntt_bench.cu:8-79).  Raw table pointers, as there: twiddles[limb * N + k] and the
// Shoup quotients in a second array, moduli as {value, const_ratio[2]} triples (DModulus, include/ntt.cuh:6-32).
//
// One workgroup per limb keeps the whole polynomial in LDS (N * 8 bytes <= 16 KiB); each thread owns N / (2 T)
// butterflies per stage, T = min(N / 2, 256).  Stage order, twiddle addressing tw[m + k] and the final
// canonicalisation are the reference's; so is the inverse's scaling of the FIRST half only (the second half gets its
// N^-1 through slot 1 of the inverse table, src/host/ntt.cu:53-55).
#include "pha_internal.h"

namespace pha {

struct Ntt1dArgs {
    u64 *inout;
    const u64 *tw, *tw_shoup;
    const DModulus *mod;
    const u64 *scalar, *scalar_shoup;   // inverse only, indexed by limb
    uint32_t n, log_n, first_limb;
};

template <bool FWD>
__global__ __launch_bounds__(256) void ntt1d_kernel(const Ntt1dArgs k) {
    extern __shared__ u64 buf[];
    const uint32_t limb = k.first_limb + blockIdx.x, n = k.n, half = n >> 1;
    const u64 q = k.mod[limb].value, q2 = q << 1;
    u64 *data = k.inout + (size_t)limb * n;
    const u64 *tw = k.tw + (size_t)limb * n, *tws = k.tw_shoup + (size_t)limb * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) buf[i] = data[i];
    __syncthreads();
    if (FWD) {
        // m groups of n / m elements; pair distance t = n / (2 m)   (ntt_1d.cu:40-52)
        for (uint32_t m = 1, t = half; m < n; m <<= 1, t >>= 1) {
            for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
                const uint32_t g = b / t, j = b - g * t, i0 = 2 * g * t + j;
                u64 x = buf[i0], y = buf[i0 + t];
                ct_bfly(x, y, u64x2{tw[m + g], tws[m + g]}, q, q2);
                buf[i0] = x;
                buf[i0 + t] = y;
            }
            __syncthreads();
        }
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) data[i] = csub(csub(buf[i], q2), q);   // :56-61
    } else {
        for (uint32_t m = half, t = 1; m >= 1; m >>= 1, t <<= 1) {                                     // :211-224
            for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
                const uint32_t g = b / t, j = b - g * t, i0 = 2 * g * t + j;
                u64 x = buf[i0], y = buf[i0 + t];
                gs_bfly(x, y, u64x2{tw[m + g], tws[m + g]}, q, q2);
                buf[i0] = x;
                buf[i0 + t] = y;
            }
            __syncthreads();
        }
        const u64x2 sc{k.scalar[limb], k.scalar_shoup[limb]};
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const u64 v = csub(buf[i], q);                                                             // :226-230
            data[i] = i < half ? shoup(v, sc, q) : v;                                                  // :232-235
        }
    }
}

static void launch_1d(bool fwd, u64 *inout, const u64 *tw, const u64 *tws, const u64 *modulus, const u64 *scalar,
                      const u64 *scalar_shoup, size_t dim, size_t cms, size_t start, hipStream_t s) {
    if (!inout || !tw || !tws || !modulus || (!fwd && (!scalar || !scalar_shoup)))
        throw std::invalid_argument("null device pointer");
    if (dim < 2 || dim > 2048 || (dim & (dim - 1))) throw std::invalid_argument("dim must be a power of two in [2, 2048]");
    if (cms == 0) return;
    if (cms > 65535) throw std::invalid_argument("coeff_modulus_size out of range");
    uint32_t log_n = 0;
    while ((size_t(1) << log_n) < dim) log_n++;
    Ntt1dArgs k{inout, tw, tws, reinterpret_cast<const DModulus *>(modulus), scalar, scalar_shoup, (uint32_t)dim, log_n,
                (uint32_t)start};
    const unsigned threads = (unsigned)std::min<size_t>(256, std::max<size_t>(dim / 2, 1));
    if (fwd) hipLaunchKernelGGL(ntt1d_kernel<true>, dim3((unsigned)cms), dim3(threads), dim * sizeof(u64), s, k);
    else hipLaunchKernelGGL(ntt1d_kernel<false>, dim3((unsigned)cms), dim3(threads), dim * sizeof(u64), s, k);
    check_launch();
}

}  // namespace pha

using namespace pha;

extern "C" {

int pha_fnwt_1d(uint64_t *inout, const uint64_t *twiddles, const uint64_t *twiddles_shoup, const uint64_t *modulus,
                size_t dim, size_t coeff_modulus_size, size_t start_modulus_idx, void *stream) {
    PHA_API_BEGIN
    launch_1d(true, inout, twiddles, twiddles_shoup, modulus, nullptr, nullptr, dim, coeff_modulus_size, start_modulus_idx,
              as_stream(stream));
    PHA_API_END
}

int pha_fnwt_1d_opt(uint64_t *inout, const uint64_t *twiddles, const uint64_t *twiddles_shoup, const uint64_t *modulus,
                    size_t dim, size_t coeff_modulus_size, size_t start_modulus_idx, void *stream) {
    PHA_API_BEGIN
    (void)start_modulus_idx;   // the reference's kernel takes its limb from the block index alone (ntt_1d.cu:92-93)
    if (dim < 4) throw std::invalid_argument("dim must be a power of two in [4, 2048]");   // its first and last stages are peeled
    launch_1d(true, inout, twiddles, twiddles_shoup, modulus, nullptr, nullptr, dim, coeff_modulus_size, 0, as_stream(stream));
    PHA_API_END
}

int pha_inwt_1d(uint64_t *inout, const uint64_t *itwiddles, const uint64_t *itwiddles_shoup, const uint64_t *modulus,
                const uint64_t *scalar, const uint64_t *scalar_shoup, size_t dim, size_t coeff_modulus_size,
                size_t start_modulus_idx, void *stream) {
    PHA_API_BEGIN
    launch_1d(false, inout, itwiddles, itwiddles_shoup, modulus, scalar, scalar_shoup, dim, coeff_modulus_size,
              start_modulus_idx, as_stream(stream));
    PHA_API_END
}

int pha_inwt_1d_opt(uint64_t *inout, const uint64_t *itwiddles, const uint64_t *itwiddles_shoup, const uint64_t *modulus,
                    const uint64_t *scalar, const uint64_t *scalar_shoup, size_t dim, size_t coeff_modulus_size,
                    size_t start_modulus_idx, void *stream) {
    return pha_inwt_1d(inout, itwiddles, itwiddles_shoup, modulus, scalar, scalar_shoup, dim, coeff_modulus_size,
                       start_modulus_idx, stream);   // the reference launches the same kernel for both (ntt_1d.cu:271-292)
}

}  // extern "C"
