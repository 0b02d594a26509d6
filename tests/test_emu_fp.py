"""The FP64 modular helpers of phantom-fhe_amd/csrc/pha_arith.h (host/device functions) compiled for the host and checked against
big-integer arithmetic: the exactness claims the FP64 tensor product, the light butterflies and their re-centring schedule rest
on, at the corners of their input ranges -- no GPU needed.  Harness: tests/emu/emu_fp.cpp (test-only)."""
import ctypes as C
import itertools
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from util import rng_for

HERE = os.path.dirname(os.path.abspath(__file__))
u64p = C.POINTER(C.c_uint64)
BITS = [30, 40, 42, 47, 48, 49, 50]


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libemu_fp.so")
    # -ffp-contract=off as the library has it (the header also carries the pragma: the flag is belt and braces)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, os.path.join(HERE, "emu", "emu_fp.cpp")])
    L = C.CDLL(out)
    L.emu_fp_tensor.argtypes = [C.c_uint64, C.c_size_t] + [u64p] * 5
    L.emu_fp_square.argtypes = [C.c_uint64, C.c_size_t] + [u64p] * 3
    L.emu_fp_mulmod.argtypes = [C.c_uint64, C.c_size_t, u64p, u64p, C.c_int, C.c_int, u64p]
    L.emu_fp_light_bound.argtypes = [C.c_uint64, C.c_size_t, u64p, u64p, C.c_int]
    L.emu_fp_light_bound.restype = C.c_uint64
    L.emu_redc90.argtypes = [C.c_uint64, C.c_size_t] + [u64p] * 5
    L.emu_split_acc.argtypes = [C.c_size_t, C.c_size_t] + [u64p] * 6
    return L


def p(a):
    return a.ctypes.data_as(u64p)


def corners(q):
    return sorted({0, 1, 2, 3, q - 1, q - 2, q - 3, (q - 1) // 2, (q + 1) // 2, q // 3, 2 * q // 3, (1 << 25) % q, ((1 << 25) + 1) % q,
                   (1 << 26) % q, q - (1 << 25) % q, int(q ** 0.5), int(q ** 0.5) + 1, q - int(q ** 0.5)})


def primes_of(bits):
    # the largest NTT prime of that width for N = 2^16 and, at 50 bits, the first and last data primes of the C3 set
    ps = [int(O.get_primes(1 << 16, bits, 1)[0])]
    if bits == 50:
        c3 = [int(x) for x in O.coeff_modulus_create(1 << 16, [60] + [50] * 44 + [60] * 15)]
        ps += [c3[1], c3[44]]
    return ps


@pytest.mark.parametrize("bits", BITS)
def test_tensor_and_square_equal_big_integer_arithmetic(emu, bits):
    for q in primes_of(bits):
        cs = corners(q)
        quad = np.array(list(itertools.product(cs, repeat=4)), dtype=np.uint64)   # every corner combination of (c0, c1, c0', c1')
        r = rng_for(bits)
        rnd = r.integers(0, q, (200_000, 4), dtype=np.uint64)
        x = np.ascontiguousarray(np.concatenate([quad, rnd]).T)
        out = np.zeros(3 * x.shape[1], dtype=np.uint64)
        emu.emu_fp_tensor(q, x.shape[1], p(x[0]), p(x[1]), p(x[2]), p(x[3]), p(out))
        a0, a1, b0, b1 = (x[i].astype(object) for i in range(4))
        d0, d2 = a0 * b0 % q, a1 * b1 % q
        d1 = ((a0 + a1) * (b0 + b1) - d0 - d2) % q
        got = out.reshape(-1, 3).astype(object)
        assert np.array_equal(got[:, 0], d0) and np.array_equal(got[:, 1], d1) and np.array_equal(got[:, 2], d2)
        out2 = np.zeros(3 * x.shape[1], dtype=np.uint64)
        emu.emu_fp_square(q, x.shape[1], p(x[0]), p(x[1]), p(out2))
        got = out2.reshape(-1, 3).astype(object)
        assert np.array_equal(got[:, 0], a0 * a0 % q) and np.array_equal(got[:, 1], 2 * a0 * a1 % q) and np.array_equal(got[:, 2], a1 * a1 % q)


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("shift", [-7, -4, -1, 0, 1, 3, 6])
def test_mulmod_forms_on_lazy_inputs(emu, bits, shift):
    """Y = y + shift q with y in [0, q): the lazy range the butterflies hand to a product (|Y| <= 7 q < 2^52.9 at 50 bits).  The full
    product is exact and centred everywhere; the light one is exact and obeys the bound the re-centring schedule builds on,
    |r| <= q (0.5 + 1.5 |Y| 2^-52)."""
    for q in primes_of(bits):
        cs = corners(q)
        pairs = np.array(list(itertools.product(cs, repeat=2)), dtype=np.uint64)
        r = rng_for(bits * 31 + shift + 7)
        x = np.ascontiguousarray(np.concatenate([pairs, r.integers(0, q, (300_000, 2), dtype=np.uint64)]).T)
        y, w = x[0].astype(object), x[1].astype(object)
        ref = (y + shift * q) * w % q
        for light in (0, 1):
            out = np.zeros(x.shape[1], dtype=np.uint64)
            emu.emu_fp_mulmod(q, x.shape[1], p(x[0]), p(x[1]), shift, light, p(out))
            assert np.array_equal(out.astype(object), ref), (q, shift, light)
        worst = emu.emu_fp_light_bound(q, x.shape[1], p(x[0]), p(x[1]), shift) / 1048576.0
        m = max(abs(shift), abs(shift + 1))            # |Y| <= m q
        assert worst <= 0.5 + 1.5 * m * q / 2.0 ** 52 + 1e-5, (q, shift, worst)


# ---- r06: mont_redc90_split, the word-wise Montgomery reduction of the base conversion's split accumulators ----------------------
def _redc90_expect(ll, lh, hl, hh, q):
    inv = pow(1 << 90, -1, q)
    v = ll.astype(object) + ((lh.astype(object) + hl.astype(object)) << 30) + (hh.astype(object) << 60)
    return v * inv % q


REDC_PRIMES = [(1 << 60) - 93, 1152921504606584833, 1125899906826241, (1 << 50) - 27, (1 << 40) - 87, (1 << 31) - 1, 1073741827, 65537, 12289, 3]


@pytest.mark.parametrize("q", REDC_PRIMES)
@pytest.mark.parametrize("terms", [1, 2, 8, 14, 15])
def test_redc90_on_conversion_sums(emu, q, terms):
    """sum_i y_i * row_i accumulated as the kernels do (30-bit halves, four 64-bit sums), y_i below 2^60 (any input prime up to
    60 bits), row_i in [0, q): the function returns sum * 2^-90 mod q, canonical -- for every modulus size the converters meet
    (60-bit special primes, 50-bit data primes, a BFV plain modulus) and every corner of the inputs."""
    r = rng_for(q % 1000003 + terms)
    count = 20_000
    top = (1 << 60) - 1
    y = r.integers(0, 1 << 60, (terms, count), dtype=np.uint64)
    y[:, 0] = top                                   # every half at its maximum
    y[:, 1] = 0
    y[:, 2] = (1 << 30) - 1
    y[:, 3] = ((1 << 30) - 1) << 30
    y[:, 4] = 1 << 30
    y = np.ascontiguousarray(y)
    for rows in ([q - 1] * terms, [1] * terms, [int(x) for x in r.integers(0, q, terms, dtype=np.uint64)],
                 [q - 1 - int(x) for x in r.integers(0, min(q, 1 << 20), terms, dtype=np.uint64)]):
        m = np.array(rows, dtype=np.uint64)
        acc = [np.zeros(count, dtype=np.uint64) for _ in range(4)]
        emu.emu_split_acc(terms, count, p(y), p(m), *[p(a) for a in acc])
        out = np.zeros(count, dtype=np.uint64)
        emu.emu_redc90(q, count, *[p(a) for a in acc], p(out))
        want = (y.astype(object) * m.astype(object)[:, None]).sum(axis=0) * pow(1 << 90, -1, q) % q
        assert np.array_equal(out.astype(object), want)
        assert np.array_equal(out.astype(object), _redc90_expect(*acc, q))


@pytest.mark.parametrize("q", [(1 << 60) - 93, 1152921504606584833, (1 << 59) + 21])
def test_redc90_at_the_accumulator_bound(emu, q):
    """The overflow argument in pha_arith.h: every accumulator at its ceiling 15 (2^30 - 1)^2 (no row below q reaches it; the
    function must not wrap there all the same), neighbours of it, and random accumulators below it, with a 60-bit modulus
    (V < 2^124 = 2^64 * 2^60 keeps the result below 2q)."""
    assert q % 2 == 1
    h2 = 15 * ((1 << 30) - 1) ** 2
    r = rng_for(7)
    cs = [h2, h2 - 1, h2 - ((1 << 30) - 1), h2 >> 1, (1 << 63) + 12345, 0, 1, (1 << 30) - 1, 1 << 30, (1 << 34) - 1]
    grid = np.array(list(itertools.product(cs, repeat=4)), dtype=np.uint64)
    rnd = r.integers(0, h2 + 1, (100_000, 4), dtype=np.uint64)
    x = np.ascontiguousarray(np.concatenate([grid, rnd]).T)
    out = np.zeros(x.shape[1], dtype=np.uint64)
    emu.emu_redc90(q, x.shape[1], p(x[0]), p(x[1]), p(x[2]), p(x[3]), p(out))
    assert np.array_equal(out.astype(object), _redc90_expect(x[0], x[1], x[2], x[3], q))
