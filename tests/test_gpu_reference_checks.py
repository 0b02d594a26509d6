"""The only pass criteria the reference's own tree holds for this path, re-stated as GPU tests through the C ABI.
They are property checks -- the reference has no golden vectors -- so they do not pin the oracle; they are what its maintainers run.

* test/ntt_test.cu:71-122 (`test_nwt_2d`): a polynomial of constants goes through nwt_2d_radix8_forward_inplace and
  nwt_2d_radix8_backward_inplace and must come back unchanged, for log N = 12 .. 17 with 50-bit primes (ntt_test.cu:78) and the
  batch sizes of its loop.
* benchmark/matmul_bench.cu:545-673 (`main`, check at :658-663): A = B = all ones, 30 moduli x 256^3, every entry of C must equal k.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
def test_nwt_2d_round_trip_on_constants(log_n, gpu):
    """test/ntt_test.cu:71-122: `h_idata[i] = const`, forward, backward, compare with the input; here for every constant of a small
    set (0, 1, the ntt_test value 2, q - 1) and for 1, 2 and 8 limbs per launch, plus the out-of-place inverse."""
    import phantom_fhe_amd as P
    n = 1 << log_n
    limbs = 8
    primes = [int(p) for p in P.coeff_modulus_create(n, [50] * limbs)]          # ntt_test.cu:78: CoeffModulus::Create(dim, {50, ...})
    ctx = P.PhantomContext(log_n, primes, 0, device=gpu)
    for const in (0, 1, 2, "q-1"):
        x = np.stack([np.full(n, (q - 1) if const == "q-1" else const, dtype=np.uint64) for q in primes])
        for count in (1, 2, limbs):
            d = P.to_device(x[:count], gpu)
            ctx.nwt_2d_radix8_forward_inplace(d, count, 0)
            fwd = P.to_host(d)
            if const == 0:
                assert not fwd.any()                                              # the transform of zero is zero
            elif log_n <= 14:
                assert all(fwd[i].max() < primes[i] for i in range(count))        # canonical outputs (fntt_2d.cu:187-193)
            ctx.nwt_2d_radix8_backward_inplace(d, count, 0)
            assert np.array_equal(P.to_host(d), x[:count]), (log_n, const, count)
            d_out = P.to_device(np.zeros_like(x[:count]), gpu)
            ctx.nwt_2d_radix8_backward(d_out, P.to_device(fwd, gpu), count, 0)      # intt_2d.cu:724-757 out-of-place form
            assert np.array_equal(P.to_host(d_out), x[:count])


def test_matmul_bench_all_ones_gives_k(gpu):
    """benchmark/matmul_bench.cu:658-663: with A and B all ones every entry of every C_z is k mod q_z = 256, at the benchmark's own
    shape (30 moduli x 256 x 256 x 256, 50-bit moduli), through pha_batched_modular_gemm; plus a 60-bit modulus set (the wide-digit
    path) and a ragged shape."""
    import torch
    import phantom_fhe_amd as P
    for bits, batch, (m, nn, k) in ((50, 30, (256, 256, 256)), (60, 4, (256, 256, 256)), (50, 3, (96, 160, 200))):
        primes = [int(p) for p in P.coeff_modulus_create(4096, [bits] * batch)]
        ctx = P.PhantomContext(12, primes, 0, device=gpu)
        a = torch.ones((batch, m, k), dtype=torch.int64, device=gpu)
        b = torch.ones((batch, k, nn), dtype=torch.int64, device=gpu)
        c = torch.zeros((batch, m, nn), dtype=torch.int64, device=gpu)
        ctx.batched_modular_gemm(c, a, b, m, nn, k, batch)
        assert bool((c == k).all()), (bits, batch, m, nn, k)
