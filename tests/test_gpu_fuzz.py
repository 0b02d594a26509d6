"""Seeded random parameter sets through the whole key-switch path on the GPU vs the oracle: mixed prime sizes
(30..60 bits: FP64 light / FP64 / integer NTT paths side by side), alpha 1..4, dnum 1..4, every level including
short last digits, all three schemes, plus rescale / modulus switch and hoisted rotations."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from util import rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _chain(rng, n, count):
    """`count` distinct NTT primes with random bit sizes (largest first within a size, like CoeffModulus::Create)."""
    sizes = [int(rng.choice([30, 36, 40, 42, 43, 45, 47, 48, 50, 55, 60])) for _ in range(count)]
    pools = {b: [int(p) for p in O.get_primes(n, b, sizes.count(b))] for b in set(sizes)}
    return [pools[b].pop(0) for b in sizes]


# PHA_FUZZ_EXTRA=k adds k more seeded cases (a longer hunt, not part of the default run)
@pytest.mark.parametrize("case", list(range(27)) + [100 + i for i in range(int(os.environ.get("PHA_FUZZ_EXTRA", "0")))])
def test_random_parameter_sets(case, gpu):
    import phantom_fhe_amd as P
    rng = rng_for(5000 + case)
    log_n = int(rng.choice([12, 12, 13]))
    alpha = int(rng.integers(1, 5))
    dnum = int(rng.integers(1, 5))
    if case >= 24:                      # the largest degrees once per scheme (small chains keep the oracle quick)
        log_n, alpha, dnum = (17, 2, 2) if case < 26 else (16, 3, 1)
    n = 1 << log_n
    size_q = alpha * dnum
    primes = _chain(rng, n, size_q + alpha)
    scheme = [O.CKKS, O.BFV, O.BGV][case % 3]
    ql = int(rng.integers(1, size_q + 1))
    oc = O.Ctx(log_n, primes, alpha)
    ctx = P.PhantomContext(log_n, primes, alpha, device=gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        t = 65537
        ctx.set_plain_modulus(t)
        tool.set_plain_modulus(t)
    evk = np.stack([np.stack([uniform_poly(rng, primes, n), uniform_poly(rng, primes, n)]) for _ in range(dnum)])
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    ct = np.stack([uniform_poly(rng, primes[:ql], n) for _ in range(2)])
    c2 = uniform_poly(rng, primes[:ql], n)
    d_ct = P.to_device(ct, gpu)
    ctx.keyswitch_inplace(ql, d_ct, P.to_device(c2, gpu), rlk.public_keys_ptr, scheme)
    ref = tool.keyswitch_inplace(ct, c2, [evk[i] for i in range(tool.beta)], scheme)
    assert np.array_equal(P.to_host(d_ct), ref), f"keyswitch {primes} alpha={alpha} ql={ql} scheme={scheme}"
    # batch of two through the batched entry point: same answers
    d_b = P.to_device(np.stack([ct, ref]), gpu)
    d_c2 = P.to_device(np.stack([c2, ct[0]]), gpu)
    ctx.keyswitch_inplace_batched(ql, d_b, d_c2, 2, rlk.public_keys_ptr, scheme)
    got = P.to_host(d_b)
    assert np.array_equal(got[0], ref)
    assert np.array_equal(got[1], tool.keyswitch_inplace(ref, ct[0], [evk[i] for i in range(tool.beta)], scheme))
    if ql > 1:
        dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
        if scheme == O.CKKS:
            ctx.divide_and_round_q_last_ntt(ql, P.to_device(ref, gpu), 2, dst)
            assert np.array_equal(P.to_host(dst), tool.rescale_ntt(ref, 2))
        elif scheme == O.BGV:
            ctx.mod_t_and_divide_q_last_ntt(ql, P.to_device(ref, gpu), 2, dst)
            assert np.array_equal(P.to_host(dst), tool.mod_t_divide_q_last_ntt(ref, 2))
        else:
            ctx.divide_and_round_q_last(ql, P.to_device(ref, gpu), 2, dst)
            assert np.array_equal(P.to_host(dst), tool.divide_and_round_q_last(ref, 2))
    elts = [5, 2 * n - 1]
    glk = [np.stack([np.stack([uniform_poly(rng, primes, n), uniform_poly(rng, primes, n)]) for _ in range(dnum)]) for _ in elts]
    d_glk = [P.PhantomRelinKey.from_numpy(k, gpu) for k in glk]
    o_glk = [[k[i] for i in range(tool.beta)] for k in glk]
    d_h = P.to_device(ct, gpu)
    ctx.hoisting(ql, d_h, elts, d_glk, scheme)
    assert np.array_equal(P.to_host(d_h), tool.hoisting(ct, elts, o_glk, scheme))
    keys = [evk[i] for i in range(tool.beta)]
    # r03 entries.  key switch + rescale in one call (ckks): the two reference steps, bit for bit, alone and as a batch of two
    if scheme == O.CKKS and ql > 1:
        want = tool.rescale_ntt(ref, 2)
        fused = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
        ctx.keyswitch_rescale(ql, P.to_device(ct, gpu), P.to_device(c2, gpu), rlk.public_keys_ptr, fused)
        assert np.array_equal(P.to_host(fused), want), f"keyswitch_rescale {primes} alpha={alpha} ql={ql}"
        fused2 = P.to_device(np.zeros((2, 2, ql - 1, n), dtype=np.uint64), gpu)
        ctx.keyswitch_rescale_batched(ql, P.to_device(np.stack([ct, ref]), gpu), P.to_device(np.stack([c2, ct[0]]), gpu), 2,
                                      rlk.public_keys_ptr, fused2)
        got2 = P.to_host(fused2)
        assert np.array_equal(got2[0], want)
        assert np.array_equal(got2[1], tool.rescale_ntt(tool.keyswitch_inplace(ref, ct[0], keys, scheme), 2))
    # relinearize + rotate of a batch in one call (BASELINE config 4's composition) against the oracle's three steps
    from phantom_fhe_amd import workloads as W
    ct3 = np.stack([np.concatenate([ct, c2[None]]), np.concatenate([ref, ct[1][None]])])
    rr = P.to_host(W.relinearize_rotate_batch(ctx, ql, P.to_device(ct3, gpu), rlk, d_glk[0], elts[0], scheme))
    table = O.galois_ntt_table(log_n, elts[0])
    for b in range(2):
        x = tool.keyswitch_inplace(ct3[b, :2], ct3[b, 2], keys, scheme)
        if scheme == O.BFV:
            g = [oc.apply_galois_coeff(x[p], elts[0], ql) for p in range(2)]
        else:
            g = [O.apply_galois_ntt(x[p], table, n, ql) for p in range(2)]
        assert np.array_equal(rr[b], tool.keyswitch_inplace(np.stack([g[0], np.zeros_like(g[0])]), g[1], o_glk[0], scheme)), b
    # baby-step / giant-step weighted hoisting (BASELINE config 5's composition), NTT-domain schemes
    if scheme != O.BFV:
        qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(alpha)]]
        ws = [[uniform_poly(rng, qlp_primes, n) for _ in range(2)] for _ in range(2)]
        d_ws = [[P.to_device(w, gpu) for w in row] for row in ws]
        bs = P.to_host(W.diag_matvec_bsgs(ctx, ql, P.to_device(ct, gpu), [1, elts[0]], [None, d_glk[0]], [1, elts[1]], [None, d_glk[1]],
                                          d_ws, scheme))
        assert np.array_equal(bs, tool.hoisting_weighted_bsgs(ct, [1, elts[0]], [None, o_glk[0]], [1, elts[1]], [None, o_glk[1]], ws, scheme))


@pytest.mark.parametrize("case", list(range(6)) + [100 + i for i in range(int(os.environ.get("PHA_FUZZ_EXTRA", "0")) // 4)])
def test_random_bfv_multiply_parameter_sets(case, gpu):
    """Seeded random chains (uniform prime size per chain as the HPS variants need, 36..60 bits, 2..6 data primes, random
    plain modulus) through the four BFV multiply variants, the leveled primitives and the plaintext operations."""
    import phantom_fhe_amd as P
    rng = rng_for(7000 + case)
    log_n = int(rng.choice([12, 13]))
    n = 1 << log_n
    bits = int(rng.choice([36, 40, 45, 50, 55, 60]))
    size_q = int(rng.integers(2, 7))
    alpha = int(rng.choice([a for a in (1, 2, 3) if size_q % a == 0]))
    primes = [int(p) for p in O.coeff_modulus_create(n, [bits] * size_q + [60] * alpha)]
    t = int(rng.choice([65537, 786433, 1032193, 1 << 16, 1 << 20]))
    if t >= min(primes):
        t = 65537
    oc = O.Ctx(log_n, primes, alpha)
    ctx = P.PhantomContext(log_n, primes, alpha, device=gpu)
    ctx.set_plain_modulus(t)
    ct1 = np.stack([uniform_poly(rng, primes[:size_q], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(rng, primes[:size_q], n) for _ in range(2)])
    d1, d2 = P.to_device(ct1, gpu), P.to_device(ct2, gpu)
    dst = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    ctx.bfv_multiply_behz(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.Behz(oc, t).multiply(ct1, ct2)), f"behz {primes} t={t}"
    ctx.bfv_multiply_hps(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.Hps(oc, t).multiply(ct1, ct2)), f"hps {primes} t={t}"
    ctx.bfv_multiply_hps_overq(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.HpsOverQ(oc, t).multiply(ct1, ct2)), f"overq {primes} t={t}"
    ql = int(rng.integers(1, size_q))
    hq = O.HpsOverQ(oc, t, ql)
    ctx.bfv_multiply_hps_overq_leveled(ql, d1, d2, dst)
    assert np.array_equal(P.to_host(dst), hq.multiply(ct1, ct2)), f"leveled {primes} t={t} ql={ql}"
    m = rng.integers(0, t, n, dtype=np.uint64)
    d = P.to_device(ct1[0], gpu)
    ctx.bfv_add_plain(size_q, d, P.to_device(m, gpu))
    assert np.array_equal(P.to_host(d), oc.bfv_add_plain(ct1[0], m, t))
    d = P.to_device(ct1, gpu)
    ctx.bfv_multiply_plain(size_q, d, 2, P.to_device(m, gpu))
    assert np.array_equal(P.to_host(d), oc.bfv_multiply_plain(ct1, m, t))
