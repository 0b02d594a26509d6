"""CPU checks of the oracle's r06 restatements (no GPU): the launchers of include/rns.cuh:159-200 and include/rns_bconv.cuh:64-68 that
the product now exports one by one.  Each is held against something that is NOT the restatement itself: a relation between two
reference launchers, the definition in Python integers, or the exact value the operation approximates."""
import numpy as np
import pytest

from oracle import oracle as O
from util import crt_compose, oracle_ctx, primes_of, rng_for, uniform_poly


@pytest.mark.parametrize("name,ql", [("hyb12_a2", 6), ("hyb12_a2", 3), ("c1_bfv4096", 2), ("hyb13_a3", 7)])
def test_moddown_bfv_equals_moddown_from_ntt_of_the_transformed_input(name, ql):
    """DRNSTool::moddown (rns_bconv.cu:712-761) takes BFV input in coefficient form, moddown_from_NTT (:776-828) transforms it
    first: moddown(x) == moddown_from_NTT(NTT(x)); for CKKS the two store the same words on the same input."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql)
    qlp = list(primes[:ql]) + list(primes[size_q:])
    idx = list(range(ql)) + list(range(size_q, size_q + size_p))
    cx = uniform_poly(rng_for(3000 + ql), qlp, n)
    down = tool.moddown(cx, O.BFV)
    assert np.array_equal(down, tool.moddown_from_ntt(oc.nwt_forward_map(cx, idx), O.BFV))
    assert np.array_equal(tool.moddown(cx, O.CKKS), tool.moddown_from_ntt(cx, O.CKKS))
    # what it computes: (cx_j - [cx_P]_{q_j}) / P up to the conversion's small overflow: P * out + conv(cx_P) == cx_j (mod q_j)
    big_p = 1
    for p in primes[size_q:]:
        big_p *= int(p)
    conv = O.bconv(primes[size_q:], primes[:ql], cx[ql:], n)
    for j in range(ql):
        q = int(primes[j])
        for k in range(0, n, 701):
            assert (int(down[j, k]) * big_p + int(conv[j, k])) % q == int(cx[j, k])


def test_bconv_behz_var1_is_its_definition():
    """bConv_BEHZ_var1 (rns_bconv.cu:231-246; constants host/rns.cu:469-496) in Python integers."""
    log_n, primes, _ = primes_of("hyb13_a3")
    n = 64
    ip, op = [int(p) for p in primes[:4]], [int(p) for p in primes[9:12]]
    src = uniform_poly(rng_for(3100), ip, n)
    got = O.bconv_behz_var1(ip, op, src, n)
    big_p = 1
    for p in op:
        big_p *= p
    for k in range(n):
        y = []
        for i, qi in enumerate(ip):
            hat = 1
            for m, qm in enumerate(ip):
                if m != i:
                    hat = hat * qm % qi
            y.append(int(src[i, k]) * ((-big_p * pow(hat, -1, qi)) % qi) % qi)
        for j, pj in enumerate(op):
            assert int(got[j, k]) == sum(yi * pow(qi, -1, pj) for yi, qi in zip(y, ip)) % pj


@pytest.mark.parametrize("t", [65537, 1 << 20, (1 << 60) + 33])
def test_exact_convert_array_is_the_centred_lift_modulo_t(t):
    """exact_convert_array (rns_bconv.cu:374-431): the value represented by the residues, taken in (-Q/2, Q/2], modulo t."""
    log_n, primes, _ = primes_of("hyb12_a2")
    n = 256
    ip = [int(p) for p in primes[:3]]
    src = uniform_poly(rng_for(3200), ip, n)
    src[:, 0] = 0
    src[:, 1] = np.array(ip, dtype=np.uint64) - 1          # the value -1
    got = O.exact_convert_array(ip, t, src, n)
    Q = ip[0] * ip[1] * ip[2]
    for k in range(n):
        x, _ = crt_compose([src[i, k] for i in range(3)], ip)
        x = x - Q if 2 * x > Q else x
        assert int(got[k]) == x % t
    assert int(got[0]) == 0 and int(got[1]) == t - 1


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("hyb12_a2", 1032193)])
def test_behz_steps_compose_to_the_multiply_and_mean_what_they_say(name, plain_t):
    """fastbconv_m_tilde + sm_mrq lift x from q to Bsk as x + (small) * q; fast_floor + fastbconv_sk bring floor(x / q) back: the
    value semantics of the four steps (BEHZ16), on top of the whole-multiply tests that now run through them."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    behz = O.Behz(oc, plain_t)
    qp = [int(p) for p in primes[:size_q]]
    Q = 1
    for q in qp:
        Q *= q
    x = uniform_poly(rng_for(3300), qp, n)
    lifted = behz.sm_mrq(behz.fastbconv_m_tilde(x))
    for k in range(0, n, 173):
        v, _ = crt_compose([x[i, k] for i in range(size_q)], qp)
        # lifted == v + e * Q (mod every Bsk prime) for one small integer e (|e| <= size_q, the m_tilde correction leaves it tiny)
        es = set()
        for j, p in enumerate(behz.bsk):
            d = (int(lifted[j, k]) - v) * pow(Q, -1, p) % p
            es.add(d if d < p // 2 else d - p)
        assert len(es) == 1 and abs(es.pop()) <= 1
