"""Shared helpers for the test-suite: parameter sets of BASELINE.json's configs and seeded inputs."""
import functools

import numpy as np

from oracle import oracle as O

# BFVDefault(4096) literal primes, src/host/globals.cu:71 (2 data primes + 1 special)
C1_PRIMES = [0xffffee001, 0xffffc4001, 0x1ffffe0001]

CONFIGS = {
    # name: (log_n, bit sizes of QP, special_modulus_size)
    "c1_bfv4096": (12, None, 1),
    "c2_ntt14": (14, [50] * 8, 0),                       # test/ntt_test.cu:78 (50-bit primes), 8 limbs
    "hyb12_a2": (12, [60, 40, 40, 40, 40, 40, 60, 60], 2),  # 6 data limbs, alpha 2 -> beta 3
    "hyb13_a3": (13, [60] + [50] * 8 + [60] * 3, 3),  # 9 data limbs (dnum 3); lower levels have a short last digit
    "c3_ckks16": (16, [60] + [50] * 44 + [60] * 15, 15),  # examples/3_ckks.cu:729-739
    "c4_bfv15": (15, [60] + [50] * 29 + [60] * 15, 15),   # benchmark/keyswitch_bench.cu:25-34
    "c2_ckks14": (14, [60] + [40] * 7 + [60], 1),         # benchmark/ckks_bench.cu:255 (8 data limbs + 1 special)
    "hyb14_a4": (14, [60] + [50] * 7 + [60] * 4, 4),      # 8 data limbs in 2 digits of 4
    "hyb14_a2": (14, [60] + [50] * 7 + [60] * 2, 2),      # beta = 4 at N = 2^14: the one shape whose fused key products take the full (not the light) FP64 form
    "hyb17_a2": (17, [60, 50, 50, 50, 60, 60], 2),        # N = 2^17: the 256 x 512 plan under the fused mod-up + inner product (4 data limbs, beta 2)
    "hyb16_a12": (16, [60] + [50] * 23 + [60] * 12, 12),  # N = 2^16 with alpha = 12 (beta = 2): the 16-input instantiation of the r05 fused mod-up conversion
    "wide_p33": (12, [36] * 6 + [37] * 33, 33),            # special base wider than the register-resident converters (alpha > 32)
    "wide_p20": (12, [45] * 24 + [46] * 20, 20),           # 17..32 special primes: the 30 / 31-bit split converter (two digits: 20 + 4 limbs)
    "p61_a2": (12, [50] * 6 + [61, 61], 2),               # 61-bit special primes: 30 / 31 cuts in mod-up, 31 / 30 in mod-down
    "bfv13_50": (13, [50] * 4 + [60, 60], 2),             # uniform data primes: what the HPS variant of BFV multiply needs
}


@functools.lru_cache(maxsize=None)
def primes_of(name):
    log_n, bits, size_p = CONFIGS[name]
    if bits is None:
        return log_n, tuple(C1_PRIMES), size_p
    return log_n, tuple(int(x) for x in O.coeff_modulus_create(1 << log_n, list(bits))), size_p


@functools.lru_cache(maxsize=4)
def oracle_ctx(name):
    log_n, primes, size_p = primes_of(name)
    return O.Ctx(log_n, list(primes), size_p)


def uniform_poly(rng, primes, n):
    """[len(primes)][n] uniform residues, limb i in [0, primes[i])."""
    out = np.empty((len(primes), n), dtype=np.uint64)
    for i, q in enumerate(primes):
        out[i] = rng.integers(0, int(q), n, dtype=np.uint64)
    return out


def rng_for(config_id):
    return np.random.default_rng(0x5EED0000 + config_id)


def brev(k, bits):
    return int(bin(k)[2:].zfill(bits)[::-1], 2) if bits else 0


def crt_compose(residues, primes):
    """Python big-int CRT: residues[i] mod primes[i] -> value in [0, prod)."""
    M = 1
    for p in primes:
        M *= int(p)
    x = 0
    for r, p in zip(residues, primes):
        p = int(p)
        Mi = M // p
        x += int(r) * Mi * pow(Mi, -1, p)
    return x % M, M
