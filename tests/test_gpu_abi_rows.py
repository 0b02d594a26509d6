"""GPU parity of the two boundary rows that had no direct test, through the C ABI, bit-exact:

  * pha_add_to_ct                            -- add_to_ct_kernel, include/rns_bconv.cuh:217, src/rns_bconv.cu:763-769
  * pha_nwt_2d_radix8_backward_inplace_scale -- include/ntt.cuh:217, src/ntt/intt_2d.cu:759-794 (kernel :209-311)

Both at the C3 set (N = 2^16, 45 + 15 limbs) and at hyb12_a2 (N = 4096: the one-launch plan); the second with a scale vector
that differs per limb, with a non-zero start_modulus_idx, and on the special limbs (60-bit primes, the integer back end)."""
import numpy as np
import pytest

from oracle import oracle as O
from util import oracle_ctx, primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _ctx(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    return P.PhantomContext(log_n, list(primes), size_p, device=gpu)


@pytest.mark.parametrize("name,size_ql", [("hyb12_a2", 6), ("hyb12_a2", 3), ("c3_ckks16", 45), ("c3_ckks16", 31)])
def test_add_to_ct(name, size_ql, gpu):
    """ct[j] = ct[j] + cx[j] mod q_j for the size_Ql data limbs; cx is the [QlP][N] buffer of a key switch (its P limbs,
    which lie behind the first size_Ql limbs, are not touched), and is left as it was."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(70 + size_ql)
    qlp = list(primes[:size_ql]) + list(primes[len(primes) - size_p:])
    ct = uniform_poly(r, primes[:size_ql], n)
    cx = uniform_poly(r, qlp, n)
    # extremes: q - 1 + q - 1 and 0 + 0 in every limb
    for j, q in enumerate(primes[:size_ql]):
        ct[j, 0] = cx[j, 0] = int(q) - 1
        ct[j, 1] = cx[j, 1] = 0
    d_ct, d_cx = P.to_device(ct, gpu), P.to_device(cx, gpu)
    ctx.add_to_ct(d_ct, d_cx, size_ql)
    want = oc.add(ct, cx[:size_ql], size_ql, 0)
    assert np.array_equal(P.to_host(d_ct), want)
    assert np.array_equal(P.to_host(d_cx), cx)
    # the definition itself, independently of the oracle
    for j in (0, size_ql - 1):
        q = int(primes[j])
        assert np.array_equal(want[j], ((ct[j].astype(object) + cx[j].astype(object)) % q).astype(np.uint64))


@pytest.mark.parametrize("name,cms,start", [("hyb12_a2", 6, 0), ("hyb12_a2", 4, 2), ("hyb12_a2", 2, 6),
                                            ("c3_ckks16", 45, 0), ("c3_ckks16", 9, 36), ("c3_ckks16", 15, 45),
                                            ("c2_ntt14", 8, 0)])
def test_backward_inplace_scale(name, cms, start, gpu):
    """inout[j] <- iNTT(inout[j]) * scale[j] mod q_j for start <= j < start + cms: scale / scale_shoup are indexed by the limb's
    position in the buffer, as the reference's kernel indexes them (twr_idx = i / (n / 8) + start_mod_idx, scale[twr_idx]:
    intt_2d.cu:236,305-309), so the entries below `start` are never read (poisoned here); limbs outside the range are
    untouched.  scale differs per limb and includes 1, q - 1 and a small value."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    total = len(primes)
    r = rng_for(80 + start)
    x = uniform_poly(r, primes, n)
    sel = primes[start:start + cms]
    scale = np.array([r.integers(1, int(q)) for q in sel], dtype=np.uint64)
    scale[0] = 1
    scale[-1] = int(sel[-1]) - 1
    if cms > 2:
        scale[1] = 3
    shoup = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(scale, sel)], dtype=np.uint64)
    d = P.to_device(x, gpu)
    poison = np.full(start, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    ctx.nwt_2d_radix8_backward_inplace_scale(d, cms, start, P.to_device(np.concatenate([poison, scale]), gpu),
                                             P.to_device(np.concatenate([poison, shoup]), gpu))
    got = P.to_host(d)
    inv = oc.nwt_backward(x[start:start + cms], cms, start)
    want = x.copy()
    want[start:start + cms] = oc.multiply_scalar(inv, scale, cms, start)
    assert np.array_equal(got, want)
    # scale = 1 leaves the plain inverse; scale = q - 1 its negation
    assert np.array_equal(got[start], inv[0])
    assert np.array_equal(got[start + cms - 1], oc.negate(inv[cms - 1:cms], 1, start + cms - 1)[0])
    assert total >= start + cms
