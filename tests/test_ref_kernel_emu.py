"""The oracle's textbook stage loops against a thread-synchronous emulation of the reference's OWN kernels (oracle/ref_kernel_emu.py restates
inplace_fnwt_radix8_phase1 / phase2 and inplace_inwt_radix8_phase1 / phase2 with every index expression, shared-memory slot, barrier section
and lazy wrap-around butterfly of src/ntt/fntt_2d.cu:9-198, src/ntt/intt_2d.cu:9-207, include/butterfly.cuh:10-108): word for word equal
for N = 2^12 .. 2^17 -- n1 in {64, 128, 256} x n2 in {64, 128, 256, 512}, i.e. every remain_iters / tail case of the four kernels --
on the tables of src/host/ntt.cu:11-56.  SURVEY.md 8(c) / Appendix D.2 asked for this check as a committed test.  A restatement of the
kernels, not an execution of them: parity stays unpinned against an executed reference; what this pins is the restatement's STRUCTURE."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import ref_kernel_emu as E
from util import rng_for


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize("bits", [40, 50, 60, 61])
def test_reference_kernels_equal_the_textbook_transform(log_n, bits):
    n = 1 << log_n
    q = int(O.get_primes(n, bits, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)      # itw[1] carries N^-1, as src/host/ntt.cu:53-55 leaves it
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(9000 + log_n * 100 + bits)
    cases = [r.integers(0, q, n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64)]
    if log_n <= 13:
        e = np.zeros(n, dtype=np.uint64)
        e[1] = 1                                               # X -> the powers psi^(2 brev(k) + 1): the definition of the output order
        cases.append(e)
    for x in cases:
        want_f = c.nwt_forward(x.reshape(1, n), 1)[0]
        got_f = E.nwt_2d_radix8_forward_inplace(x, tw, tws, q)
        assert np.array_equal(got_f, want_f), ("forward", log_n, bits)
        assert int(got_f.max()) < q                            # canonical (fntt_2d.cu:187-193)
        want_i = c.nwt_backward(x.reshape(1, n), 1)[0]
        got_i = E.nwt_2d_radix8_backward_inplace(x, itw, itws, ni, nis, q)
        assert np.array_equal(got_i, want_i), ("inverse", log_n, bits)
        assert np.array_equal(E.nwt_2d_radix8_backward_inplace(got_f, itw, itws, ni, nis, q), x)   # test/ntt_test.cu:71-122: the round trip


def test_output_order_is_bit_reversed_evaluation():
    """out[k] = x(psi^(2 brev(k) + 1)) from the emulated kernels, against direct evaluation (N = 4096, 64 sampled k)."""
    log_n, n = 12, 4096
    q = int(O.get_primes(n, 50, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    psi = int(tw[1 << (log_n - 1)])                            # tw[brev(1)] = psi^1  (src/host/ntt.cu:27-30)
    assert pow(psi, 2 * n, q) == 1 and pow(psi, n, q) == q - 1
    r = rng_for(9100)
    x = r.integers(0, q, n, dtype=np.uint64)
    got = E.nwt_2d_radix8_forward_inplace(x, tw, tws, q)
    xs = [int(v) for v in x]
    for k in r.integers(0, n, 64):
        k = int(k)
        e = 2 * int(bin(k)[2:].zfill(log_n)[::-1], 2) + 1
        root = pow(psi, e, q)
        acc = 0
        for v in reversed(xs):                                 # Horner
            acc = (acc * root + v) % q
        assert int(got[k]) == acc, k
