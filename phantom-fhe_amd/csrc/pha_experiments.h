/* pha_experiments.h -- entry points that exist ONLY in the test-only library libphantom_amd_exp.so (every source compiled with
 * -DPHA_EXPERIMENTS).  The product library libphantom_amd.so does not export them: its kernel selection is a fixed function of
 * the launch shape (pha_ntt.hip: choose_plan).
 *
 * pha_set_tuning: key 0 = NTT variant bits (pha_ntt.hip, top of file); key 1 = base-conversion MAC (1: carry-free split
 * accumulators, 0: 128-bit carry chain); key 2 = limb-polynomials per launch from which N = 2^14 takes its one-workgroup plan;
 * keys 3 / 4 / 5 = lag, minimum tiles and split form of the one-launch transform.  Results are identical for every setting
 * (tests/test_gpu_ntt_variants.py).  The knobs are process-global: a development aid, never part of the product. */
#ifndef PHA_EXPERIMENTS_H
#define PHA_EXPERIMENTS_H
#if defined(PHA_EXPERIMENTS)
#ifdef __cplusplus
extern "C" {
#endif
int pha_set_tuning(int key, int value);
#ifdef __cplusplus
}
#endif
#endif
#endif
