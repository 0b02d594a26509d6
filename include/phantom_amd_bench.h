/* phantom_amd_bench.h -- measurement hooks of libphantom_amd.so used by bench.py and tools/ only.
 *
 * Not part of the drop-in boundary (include/phantom_amd.h): no reference launcher corresponds to them.  They exist so that a
 * timed region can hold nothing but kernel launches of the product path. */
#ifndef PHANTOM_AMD_BENCH_H
#define PHANTOM_AMD_BENCH_H
#include "phantom_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* enqueue `repeats` back-to-back forward transforms of a batch of polynomials (pha_nwt_2d_radix8_forward_inplace_batched
 * `repeats` times) from C, so that a timed region of K steps holds the 2 K kernel launches and no per-step host work */
int pha_repeat_forward_ntt_batched(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size, size_t start_modulus_idx,
                                   size_t batch, size_t poly_stride, int repeats, void *stream);

/* time `iters` back-to-back launches of the forward NTT with hipEvents on `stream`; average milliseconds per launch in *ms_out */
int pha_time_forward_ntt(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size, int iters,
                         void *stream, float *ms_out);

/* streaming calibration (r04; the in-library form of tools/stream_calib.hip): `iters` launches over `bytes` bytes (a multiple of 16),
 * one 16-byte word per lane, one trip per thread, timed with hipEvents on `stream`.  mode 0 = copy src -> dst, 1 = read-only (src),
 * 2 = write-only (dst), 3 = in-place read-modify-write of dst (what an in-place NTT pass does); nontemporal != 0 streams with the
 * `nt` policy.  *bytes_per_s = bytes read + bytes written per second: the calibrated counterparts of the nominal 8 TB/s that the
 * roofline object quotes beside them. */
int pha_time_stream(uint64_t *dst, const uint64_t *src, size_t bytes, int mode, int nontemporal, int iters, void *stream,
                    double *bytes_per_s);
/* = pha_time_stream(dst, src, bytes, 0, 1, ...): the nontemporal copy */
int pha_time_stream_copy(uint64_t *dst, const uint64_t *src, size_t bytes, int iters, void *stream, double *bytes_per_s);

/* number of scratch arenas the context currently holds (one per explicit stream, one per live host thread for the per-thread and
 * null streams; a thread's arenas are released when it exits) */
int pha_context_arena_count(pha_context_t ctx, size_t *count);

#ifdef __cplusplus
}
#endif
#endif
