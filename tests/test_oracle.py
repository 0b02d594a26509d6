"""Pins the CPU oracle (oracle/) -- runs without a GPU.

The reference ships no golden vectors (test/ntt_test.cu is a round trip on constants; examples compare
decryptions with eps = 1e-3), and it cannot be compiled here (include/host/defines.h:34 needs CUDA
headers), so the oracle is "parity unpinned against an executed reference".  It is pinned by:
  * the values SURVEY.md 8(c) records from the survey's run of the reference host code,
  * the literal default primes the reference embeds (src/host/globals.cu:71),
  * the mathematical definition of every operation (all stored outputs are canonical residues).
"""
import numpy as np
import pytest

from oracle import oracle as O
from util import C1_PRIMES, brev, crt_compose, oracle_ctx, primes_of, rng_for, uniform_poly


def test_reference_kats_from_survey():
    # CoeffModulus::Create(2^14, {60, 40x6, 60})[0], its const_ratio, its minimal root
    p = O.coeff_modulus_create(1 << 14, [60] + [40] * 6 + [60])
    assert int(p[0]) == 1152921504606683137
    assert O.const_ratio(int(p[0])) == (0x27fff00, 0x10)
    assert O.minimal_primitive_root(1 << 15, int(p[0])) == 212089012217363
    # same-size primes are handed out from the back of the descending list (modulus.cu:105-109)
    found = O.get_primes(1 << 14, 60, 2)
    assert int(p[0]) == int(found[1]) and int(p[-1]) == int(found[0]) and int(found[0]) > int(found[1])


def test_reference_default_primes_are_ntt_primes():
    for q in C1_PRIMES:  # src/host/globals.cu:71
        assert O.is_prime(q) and (q - 1) % (2 * 4096) == 0


@pytest.mark.parametrize("bits", [30, 50, 60])
def test_prime_search_definition(bits):
    n = 4096
    ps = [int(x) for x in O.get_primes(n, bits, 4)]
    assert ps == sorted(ps, reverse=True)
    v, want = (1 << bits) - 2 * n + 1, []
    while len(want) < 4:          # numth.cu:207-233: descend from 2^bits - 2N + 1 in steps of 2N
        if pow(2, v - 1, v) == 1 and all(v % s for s in (3, 5, 7, 11, 13)) and O.is_prime(v):
            want.append(v)
        v -= 2 * n
    assert ps == want


def test_minimal_root_is_minimal():
    n, q = 64, int(O.get_primes(64, 20, 1)[0])
    r = O.minimal_primitive_root(2 * n, q)
    roots = [x for x in range(1, q) if pow(x, n, q) == q - 1]
    assert r == min(roots)


def test_tables_follow_ntt_cu():
    log_n, n = 6, 64
    q = int(O.get_primes(n, 30, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    psi = O.minimal_primitive_root(2 * n, q)
    ipsi = pow(psi, -1, q)
    assert ni == pow(n, -1, q) and nis == (ni << 64) // q
    for k in range(n):
        assert int(tw[k]) == pow(psi, brev(k, log_n), q) and int(tws[k]) == (int(tw[k]) << 64) // q
        want = pow(ipsi, brev(k, log_n), q)
        if k == 1:
            want = want * ni % q                  # src/host/ntt.cu:53-55
        assert int(itw[k]) == want and int(itws[k]) == (want << 64) // q


@pytest.mark.parametrize("log_n,bits", [(4, 20), (6, 30), (8, 50), (8, 60)])
def test_forward_is_evaluation_at_odd_powers(log_n, bits):
    n = 1 << log_n
    q = int(O.get_primes(n, bits, 1)[0])
    psi = O.minimal_primitive_root(2 * n, q)
    c = O.Ctx(log_n, [q], 0)
    x = rng_for(log_n).integers(0, q, n, dtype=np.uint64)
    y = c.nwt_forward(x.reshape(1, n), 1)[0]
    for k in range(n):   # out[k] = x(psi^(2*brev(k)+1)), SURVEY.md a3
        assert int(y[k]) == sum(int(x[j]) * pow(psi, (2 * brev(k, log_n) + 1) * j, q) for j in range(n)) % q
    assert np.array_equal(c.nwt_backward(y.reshape(1, n), 1)[0], x)


def test_negacyclic_convolution_and_edges():
    log_n, n = 8, 256
    q = int(O.get_primes(n, 50, 1)[0])
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(77)
    a, b = r.integers(0, q, n, dtype=np.uint64), r.integers(0, q, n, dtype=np.uint64)
    fa, fb = c.nwt_forward(a.reshape(1, n), 1), c.nwt_forward(b.reshape(1, n), 1)
    prod = c.nwt_backward(c.multiply(fa, fb, 1), 1)[0]
    want = [0] * n
    for i in range(n):
        for j in range(n):
            k, s = (i + j) % n, (-1 if i + j >= n else 1)
            want[k] = (want[k] + s * int(a[i]) * int(b[j])) % q
    assert [int(v) for v in prod] == want
    for v in (0, q - 1):   # constant / extreme inputs round-trip (ntt_test.cu uses all-ones / all-twos)
        x = np.full((1, n), v, dtype=np.uint64)
        assert np.array_equal(c.nwt_backward(c.nwt_forward(x, 1), 1), x)


def test_dyadic_against_python_ints():
    log_n, primes, size_p = primes_of("hyb12_a2")
    oc = oracle_ctx("hyb12_a2")
    n, L = 1 << log_n, 3
    r = rng_for(5)
    a, b, d = (uniform_poly(r, primes[:L], n) for _ in range(3))
    a[:, 0] = 0
    b[:, 1] = [q - 1 for q in primes[:L]]
    a[:, 1] = [q - 1 for q in primes[:L]]
    for l, q in enumerate(primes[:L]):
        ai, bi, di = ([int(v) for v in arr[l][:64]] for arr in (a, b, d))
        assert [int(v) for v in oc.add(a, b, L)[l][:64]] == [(x + y) % q for x, y in zip(ai, bi)]
        assert [int(v) for v in oc.sub(a, b, L)[l][:64]] == [(x - y) % q for x, y in zip(ai, bi)]
        assert [int(v) for v in oc.negate(a, L)[l][:64]] == [(-x) % q for x in ai]
        assert [int(v) for v in oc.multiply(a, b, L)[l][:64]] == [x * y % q for x, y in zip(ai, bi)]
        assert [int(v) for v in oc.multiply_and_add(a, b, d, L)[l][:64]] == [(x * y + z) % q for x, y, z in zip(ai, bi, di)]
    t = oc.tensor_prod_2x2(np.stack([a, b]), np.stack([d, a]), L)
    for l, q in enumerate(primes[:L]):
        for k in range(16):
            c0, c1, e0, e1 = int(a[l, k]), int(b[l, k]), int(d[l, k]), int(a[l, k])
            assert [int(t[p, l, k]) for p in range(3)] == [c0 * e0 % q, (c0 * e1 + c1 * e0) % q, c1 * e1 % q]
    s = oc.tensor_square_2x2(np.stack([a, b]), L)
    assert np.array_equal(s, oc.tensor_prod_2x2(np.stack([a, b]), np.stack([a, b]), L))


def test_fast_base_conversion_is_crt_plus_small_multiple_of_q():
    """bConv_BEHZ (rns_bconv.cu:212-229) is the uncorrected conversion: out_j = (x + u*Q) mod p_j, 0<=u<#ibase."""
    log_n, primes, size_p = primes_of("hyb12_a2")
    n = 1 << log_n
    ib, ob = list(primes[6:8]) + [primes[0]], list(primes[1:5])
    src = uniform_poly(rng_for(9), ib, n)
    out = O.bconv(ib, ob, src, n)
    for k in range(0, n, 257):
        x, Q = crt_compose([src[i, k] for i in range(len(ib))], ib)
        us = {next(u for u in range(len(ib) + 1) if (x + u * Q) % int(p) == int(out[j, k])) for j, p in enumerate(ob)}
        assert len(us) == 1     # one common overflow count u for all output primes


def _small_keys(oc, primes, n, size_q, size_p, r, new_key_ntt, sk_ntt):
    dnum = size_q // size_p
    a = np.stack([uniform_poly(r, primes, n) for _ in range(dnum)])
    e_small = r.integers(-3, 4, (dnum, n))
    e = np.stack([np.stack([np.array([int(v) % int(q) for v in e_small[d]], dtype=np.uint64) for q in primes]) for d in range(dnum)])
    e_ntt = np.stack([oc.nwt_forward(e[d], len(primes), 0) for d in range(dnum)])
    return oc.gen_kswitch_key(sk_ntt, new_key_ntt, a, e_ntt)


@pytest.mark.parametrize("name,ql", [("hyb12_a2", 6), ("hyb12_a2", 3), ("c1_bfv4096", 2)])
def test_relinearize_decrypts_correctly(name, ql):
    """Functional pin of mod-up / inner product / mod-down and of the evk layout (secretkey.cu:297-341):
    with genuine keys, c0' + c1' s = c0 + c1 s + c2 s^2 + small, checked with big integers (CKKS form)."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    r = rng_for(31)
    s_small = r.integers(-1, 2, n)
    sk = np.stack([np.array([int(v) % int(q) for v in s_small], dtype=np.uint64) for q in primes])
    sk_ntt = oc.nwt_forward(sk, len(primes), 0)
    s2_ntt = oc.multiply(sk_ntt[:size_q], sk_ntt[:size_q], size_q)
    evk = _small_keys(oc, primes, n, size_q, size_p, r, s2_ntt, sk_ntt)
    tool = O.Tool(oc, ql)
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(3)])    # (c0, c1, c2) in NTT form
    out = tool.keyswitch_inplace(ct[:2], ct[2], [evk[i] for i in range(tool.beta)], O.CKKS)
    s1, s2 = sk_ntt[:ql], s2_ntt[:ql]
    before = oc.add(oc.add(ct[0], oc.multiply(ct[1], s1, ql), ql), oc.multiply(ct[2], s2, ql), ql)
    after = oc.add(out[0], oc.multiply(out[1], s1, ql), ql)
    diff = oc.nwt_backward(oc.sub(after, before, ql), ql)
    Q = 1
    for q in primes[:ql]:
        Q *= int(q)
    worst = 0
    for k in range(0, n, 97):
        v, _ = crt_compose([diff[l, k] for l in range(ql)], primes[:ql])
        v = v - Q if v > Q // 2 else v
        worst = max(worst, abs(v))
    assert worst < n * 64 * (size_q // size_p + ql)   # key-switch noise, nowhere near Q
    assert worst.bit_length() < Q.bit_length() - 20


@pytest.mark.parametrize("name,ql,plain_t", [("hyb12_a2", 6, 65537), ("hyb12_a2", 4, 786433), ("hyb13_a3", 9, 65537)])
def test_bgv_relinearize_keeps_the_plaintext_mod_t(name, ql, plain_t):
    """BGV mod-down (bgv_moddown_kernel rns_bconv.cu:636-652): with genuine BGV keys (error scaled by t,
    secretkey.cu:268-273) the key-switched phase differs from the old one by t * small, i.e. mod t nothing moves."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    r = rng_for(37)
    s_small = r.integers(-1, 2, n)
    sk = np.stack([np.array([int(v) % int(q) for v in s_small], dtype=np.uint64) for q in primes])
    sk_ntt = oc.nwt_forward(sk, len(primes), 0)
    s2_ntt = oc.multiply(sk_ntt[:size_q], sk_ntt[:size_q], size_q)
    dnum = size_q // size_p
    a = np.stack([uniform_poly(r, primes, n) for _ in range(dnum)])
    e_small = r.integers(-3, 4, (dnum, n))
    e = np.stack([np.stack([np.array([int(v) * plain_t % int(q) for v in e_small[d]], dtype=np.uint64) for q in primes])
                  for d in range(dnum)])
    e_ntt = np.stack([oc.nwt_forward(e[d], len(primes), 0) for d in range(dnum)])
    evk = oc.gen_kswitch_key(sk_ntt, s2_ntt, a, e_ntt)
    tool = O.Tool(oc, ql).set_plain_modulus(plain_t)
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(3)])
    out = tool.keyswitch_inplace(ct[:2], ct[2], [evk[i] for i in range(tool.beta)], O.BGV)
    s1, s2 = sk_ntt[:ql], s2_ntt[:ql]
    before = oc.add(oc.add(ct[0], oc.multiply(ct[1], s1, ql), ql), oc.multiply(ct[2], s2, ql), ql)
    after = oc.add(out[0], oc.multiply(out[1], s1, ql), ql)
    diff = oc.nwt_backward(oc.sub(after, before, ql), ql)
    Q = 1
    for q in primes[:ql]:
        Q *= int(q)
    worst = 0
    for k in range(0, n, 97):
        v, _ = crt_compose([diff[l, k] for l in range(ql)], primes[:ql])
        v = v - Q if v > Q // 2 else v
        assert v % plain_t == 0
        worst = max(worst, abs(v))
    assert worst < plain_t * n * 64 * (dnum + ql + size_p)
    assert worst.bit_length() < Q.bit_length() - 20


@pytest.mark.parametrize("plain_t", [65537, 1032193, 1 << 20])
def test_bgv_mod_switch_is_exact_division_and_fixes_the_residue_mod_t(plain_t):
    """mod_t_and_divide_q_last_ntt (rns.cu:1186-1236): dst * q_last = c - d with d = c mod q_last, d = 0 mod t."""
    name, ql = "hyb12_a2", 4
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql).set_plain_modulus(plain_t)
    c = uniform_poly(rng_for(43), primes[:ql], n)
    c_ntt = oc.nwt_forward(c, ql, 0)
    out = oc.nwt_backward(tool.mod_t_divide_q_last_ntt(c_ntt.reshape(1, ql, n), 1)[0], ql - 1)
    q_last = int(primes[ql - 1])
    inv_t = pow(q_last, -1, plain_t)
    for k in range(0, n, 131):
        v, _ = crt_compose([c[l, k] for l in range(ql)], primes[:ql])
        last = int(c[ql - 1, k])
        num = v - last + q_last * ((last % plain_t) * inv_t % plain_t)
        assert num % q_last == 0 and (num - v) % plain_t == 0
        assert [int(out[l, k]) for l in range(ql - 1)] == [(num // q_last) % int(q) for q in primes[:ql - 1]]


def test_rescale_is_floor_division_by_q_last():
    """divide_and_round_q_last_ntt floors (rns.cu:1118 comment notwithstanding): dst = (c - [c]_qlast)/qlast."""
    name, ql = "hyb12_a2", 4
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql)
    c = uniform_poly(rng_for(41), primes[:ql], n)            # coefficient form
    c_ntt = oc.nwt_forward(c, ql, 0)
    out = oc.nwt_backward(tool.rescale_ntt(c_ntt.reshape(1, ql, n), 1)[0], ql - 1)
    assert np.array_equal(out, tool.divide_and_round_q_last(c.reshape(1, ql, n), 1)[0])
    for k in range(0, n, 131):
        v, _ = crt_compose([c[l, k] for l in range(ql)], primes[:ql])
        want = (v - int(c[ql - 1, k])) // int(primes[ql - 1])
        assert [int(out[l, k]) for l in range(ql - 1)] == [want % int(q) for q in primes[:ql - 1]]


def test_galois_ntt_permutation_matches_coefficient_automorphism():
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    x = uniform_poly(rng_for(51), primes[:2], n)
    for elt in (3, 5, 2 * n - 1):
        via_coeff = oc.nwt_forward(oc.apply_galois_coeff(x, elt, 2), 2, 0)
        via_ntt = O.apply_galois_ntt(oc.nwt_forward(x, 2, 0), O.galois_ntt_table(log_n, elt), n, 2)
        assert np.array_equal(via_coeff, via_ntt)


def small_weight(r, primes_qlp, qlp_idx, oc, n, bound=1 << 10):
    """A small integer polynomial as residues over [Q_l || P], NTT form: one weight of hoisting_weighted."""
    w = r.integers(-bound, bound + 1, n)
    res = np.stack([np.array([int(v) % int(q) for v in w], dtype=np.uint64) for q in primes_qlp])
    return oc.nwt_forward_map(res, qlp_idx)


@pytest.mark.parametrize("name,ql", [("hyb12_a2", 6), ("hyb12_a2", 3)])
def test_weighted_hoisting_is_the_weighted_sum_of_rotations(name, ql):
    """Config 5 composition (build-defined): with genuine Galois keys, phase(out) = sum_e w_e * galois_e(phase(ct))
    + small, the main diagonal (element 1) included without a key."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    r = rng_for(61)
    s_small = r.integers(-1, 2, n)
    sk = np.stack([np.array([int(v) % int(q) for v in s_small], dtype=np.uint64) for q in primes])
    sk_ntt = oc.nwt_forward(sk, len(primes), 0)
    elts = [1, 3, 25, 2 * n - 1]
    tabs = {e: O.galois_ntt_table(log_n, e) for e in elts}
    glk = [None if e == 1 else _small_keys(oc, primes, n, size_q, size_p, r, O.apply_galois_ntt(sk_ntt[:size_q], tabs[e], n, size_q), sk_ntt)
           for e in elts]
    tool = O.Tool(oc, ql)
    qlp_idx = list(range(ql)) + [size_q + i for i in range(size_p)]
    primes_qlp = [primes[i] for i in qlp_idx]
    weights = [small_weight(r, primes_qlp, qlp_idx, oc, n) for _ in elts]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    out = tool.hoisting_weighted(ct, elts, [None if k is None else [k[i] for i in range(tool.beta)] for k in glk], weights, O.CKKS)
    s1 = sk_ntt[:ql]
    phase = oc.add(ct[0], oc.multiply(ct[1], s1, ql), ql)
    want = np.zeros_like(phase)
    for e, w in zip(elts, weights):
        want = oc.add(want, oc.multiply(O.apply_galois_ntt(phase, tabs[e], n, ql), w[:ql], ql), ql)
    after = oc.add(out[0], oc.multiply(out[1], s1, ql), ql)
    diff = oc.nwt_backward(oc.sub(after, want, ql), ql)
    Q = 1
    for q in primes[:ql]:
        Q *= int(q)
    worst = 0
    for k in range(0, n, 97):
        v, _ = crt_compose([diff[l, k] for l in range(ql)], primes[:ql])
        v = v - Q if v > Q // 2 else v
        worst = max(worst, abs(v))
    assert 0 < worst < len(elts) * (1 << 10) * n * n * 64 * (size_q // size_p + ql)
    assert worst.bit_length() < Q.bit_length() - 20
    # no weights but ones, no identity: the plain hoisting_inplace sum
    one = np.ones((tool.size_qlp, n), dtype=np.uint64)       # NTT form of the constant 1
    plain = tool.hoisting(ct, elts[1:], [[k[i] for i in range(tool.beta)] for k in glk[1:]], O.CKKS)
    same = tool.hoisting_weighted(ct, elts[1:], [[k[i] for i in range(tool.beta)] for k in glk[1:]], [one] * 3, O.CKKS)
    assert np.array_equal(plain, same)


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("hyb12_a2", 65537), ("hyb12_a2", 1032193)])
def test_bfv_behz_multiply_decrypts_to_the_product(name, plain_t):
    """BFV BEHZ multiply (src/evaluate.cu:404-548) pinned by its meaning: for encryptions of m1, m2 the three-part
    result decrypts (d0 + d1 s + d2 s^2, scaled by t/Q and rounded) to m1 * m2 mod (X^N + 1, t)."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    q = [int(p) for p in primes[:size_q]]
    oc = oracle_ctx(name)
    behz = O.Behz(oc, plain_t)
    assert behz.size_bsk in (size_q + 1, size_q + 2) and all(p.bit_length() == 61 for p in behz.bsk)
    r = rng_for(77)
    Q = 1
    for p in q:
        Q *= p
    delta = Q // plain_t
    s_small = r.integers(-1, 2, n)
    sk = np.stack([(s_small % p).astype(np.uint64) for p in q])
    sk_ntt = oc.nwt_forward(sk, size_q, 0)

    def encrypt(m):
        a = uniform_poly(r, q, n)
        e = r.integers(-3, 4, n)
        dm = np.stack([np.array([(delta * int(v) + int(ev)) % p for v, ev in zip(m, e)], dtype=np.uint64) for p in q])
        a_s = oc.nwt_backward(oc.multiply(oc.nwt_forward(a, size_q, 0), sk_ntt, size_q), size_q)
        return np.stack([oc.sub(dm, a_s, size_q), a])

    m1 = r.integers(0, plain_t, n)
    m2 = r.integers(0, plain_t, n)
    d = behz.multiply(encrypt(m1), encrypt(m2))
    d_ntt = [oc.nwt_forward(d[i], size_q, 0) for i in range(3)]
    s2 = oc.multiply(sk_ntt, sk_ntt, size_q)
    phase = oc.add(oc.add(d_ntt[0], oc.multiply(d_ntt[1], sk_ntt, size_q), size_q), oc.multiply(d_ntt[2], s2, size_q), size_q)
    phase = oc.nwt_backward(phase, size_q)
    # exact m1 * m2 mod (X^N + 1) over the integers through one 60-bit NTT prime (|coefficients| < N t^2 < 2^59)
    big = int(O.get_primes(n, 60, 1)[0])
    bc = O.Ctx(log_n, [big], 0)
    pm = bc.nwt_backward(bc.multiply(bc.nwt_forward(m1.astype(np.uint64).reshape(1, n), 1, 0),
                                     bc.nwt_forward(m2.astype(np.uint64).reshape(1, n), 1, 0), 1), 1)[0]
    for k in range(0, n, 61):
        v, _ = crt_compose([phase[l, k] for l in range(size_q)], q)
        got = ((v * plain_t + Q // 2) // Q) % plain_t
        w = int(pm[k])
        w = w - big if w > big // 2 else w
        assert got == w % plain_t, k


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("bfv13_50", 65537), ("bfv13_50", 1032193)])
def test_bfv_hps_multiply_decrypts_to_the_product(name, plain_t):
    """BFV HPS multiply (mul_tech hps, src/evaluate.cu:674-818) pinned by its meaning, like the BEHZ test; the two
    variants give different ciphertexts (different rounding) that decrypt to the same product.  The base R the
    reference picks (|Q| + 1 primes below the smallest q_i, rns.cu:690-693) only covers t*N*Q when the data primes
    have about the same size, hence the uniform chains here.  The reference's scale-and-round kernel reduces its
    carry `alpha` in place across the R limbs (rns.cu:1733), which makes limbs j >= 1 wrong by a multiple of
    (r_0 mod r_j): noise of about one prime's size, far below Q/t; restated as is."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    q = [int(p) for p in primes[:size_q]]
    oc = oracle_ctx(name)
    hps = O.Hps(oc, plain_t)
    assert hps.size_r == size_q + 1 and all(p < min(q) and (p - 1) % (2 * n) == 0 for p in hps.r)
    a_ct, b_ct = np.stack([uniform_poly(rng_for(5), q, n) for _ in range(2)]), np.stack([uniform_poly(rng_for(6), q, n) for _ in range(2)])
    da, db = hps.multiply(a_ct, b_ct), O.Behz(oc, plain_t).multiply(a_ct, b_ct)
    Qall = 1
    for p in q:
        Qall *= p
    for k in range(0, n, 997):          # both variants approximate round(t/Q * tensor): they agree up to about one prime
        va, _ = crt_compose([da[2, l, k] for l in range(size_q)], q)
        vb, _ = crt_compose([db[2, l, k] for l in range(size_q)], q)
        dd = (va - vb) % Qall
        dd = dd - Qall if dd > Qall // 2 else dd
        assert abs(dd).bit_length() <= max(p.bit_length() for p in q) + 4
    r = rng_for(78)
    Q = 1
    for p in q:
        Q *= p
    delta = Q // plain_t
    s_small = r.integers(-1, 2, n)
    sk_ntt = oc.nwt_forward(np.stack([(s_small % p).astype(np.uint64) for p in q]), size_q, 0)

    def encrypt(m):
        a = uniform_poly(r, q, n)
        e = r.integers(-3, 4, n)
        dm = np.stack([np.array([(delta * int(v) + int(ev)) % p for v, ev in zip(m, e)], dtype=np.uint64) for p in q])
        a_s = oc.nwt_backward(oc.multiply(oc.nwt_forward(a, size_q, 0), sk_ntt, size_q), size_q)
        return np.stack([oc.sub(dm, a_s, size_q), a])

    m1 = r.integers(0, plain_t, n)
    m2 = r.integers(0, plain_t, n)
    d = hps.multiply(encrypt(m1), encrypt(m2))
    d_ntt = [oc.nwt_forward(d[i], size_q, 0) for i in range(3)]
    s2 = oc.multiply(sk_ntt, sk_ntt, size_q)
    phase = oc.add(oc.add(d_ntt[0], oc.multiply(d_ntt[1], sk_ntt, size_q), size_q), oc.multiply(d_ntt[2], s2, size_q), size_q)
    phase = oc.nwt_backward(phase, size_q)
    big = int(O.get_primes(n, 60, 1)[0])
    bc = O.Ctx(log_n, [big], 0)
    pm = bc.nwt_backward(bc.multiply(bc.nwt_forward(m1.astype(np.uint64).reshape(1, n), 1, 0),
                                     bc.nwt_forward(m2.astype(np.uint64).reshape(1, n), 1, 0), 1), 1)[0]
    for k in range(0, n, 61):
        v, _ = crt_compose([phase[l, k] for l in range(size_q)], q)
        got = ((v * plain_t + Q // 2) // Q) % plain_t
        w = int(pm[k])
        w = w - big if w > big // 2 else w
        assert got == w % plain_t, k


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("bfv13_50", 65537), ("bfv13_50", 1032193)])
def test_bfv_hps_overq_multiply_decrypts_to_the_product(name, plain_t):
    """BFV hps_overq multiply (src/evaluate.cu:674-818, overq branches) pinned by its meaning: the product decrypts to
    m1 * m2 mod t; base Rl has |Q| primes."""
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    q = [int(p) for p in primes[:size_q]]
    oc = oracle_ctx(name)
    hq = O.HpsOverQ(oc, plain_t)
    assert hq.size_r == size_q and hq.r == O.Hps(oc, plain_t).r[:size_q]
    r = rng_for(79)
    Q = 1
    for p in q:
        Q *= p
    delta = Q // plain_t
    s_small = r.integers(-1, 2, n)
    sk_ntt = oc.nwt_forward(np.stack([(s_small % p).astype(np.uint64) for p in q]), size_q, 0)
    s2 = oc.multiply(sk_ntt, sk_ntt, size_q)

    def encrypt(m):
        a = uniform_poly(r, q, n)
        e = r.integers(-3, 4, n)
        dm = np.stack([np.array([(delta * int(v) + int(ev)) % p for v, ev in zip(m, e)], dtype=np.uint64) for p in q])
        a_s = oc.nwt_backward(oc.multiply(oc.nwt_forward(a, size_q, 0), sk_ntt, size_q), size_q)
        return np.stack([oc.sub(dm, a_s, size_q), a])

    big = int(O.get_primes(n, 60, 1)[0])
    bc = O.Ctx(log_n, [big], 0)

    def check(d, ma, mb):
        d_ntt = [oc.nwt_forward(d[i], size_q, 0) for i in range(3)]
        phase = oc.add(oc.add(d_ntt[0], oc.multiply(d_ntt[1], sk_ntt, size_q), size_q), oc.multiply(d_ntt[2], s2, size_q), size_q)
        phase = oc.nwt_backward(phase, size_q)
        pm = bc.nwt_backward(bc.multiply(bc.nwt_forward(ma.astype(np.uint64).reshape(1, n), 1, 0),
                                         bc.nwt_forward(mb.astype(np.uint64).reshape(1, n), 1, 0), 1), 1)[0]
        for k in range(0, n, 61):
            v, _ = crt_compose([phase[l, k] for l in range(size_q)], q)
            got = ((v * plain_t + Q // 2) // Q) % plain_t
            w = int(pm[k])
            w = w - big if w > big // 2 else w
            assert got == w % plain_t, k

    m1, m2 = r.integers(0, plain_t, n), r.integers(0, plain_t, n)
    c1, c2 = encrypt(m1), encrypt(m2)
    check(hq.multiply(c1, c2), m1, m2)
    check(hq.multiply(c1, c1.copy()), m1, m1)
    # The reference's squaring shortcut (same object on both sides, evaluate.cu:720-731) lifts BOTH factors exactly and
    # still scales by t / Rl (scaleAndRound_HPS_QlRl_Ql), i.e. it returns (Q / Rl) times the wanted t / Q * c1^2: a
    # different ciphertext, which does not decrypt to m1^2.  Restated as is; callers that square pass two objects.
    sq = hq.multiply(c1, c1)
    assert not np.array_equal(sq, hq.multiply(c1, c1.copy()))
    with pytest.raises(AssertionError):
        check(sq, m1, m1)


@pytest.mark.parametrize("drop", [1, 2])
def test_bfv_hps_overq_leveled_primitives_and_multiply(drop):
    """hps_overq_leveled with levels dropped (src/rns.cu:897-975, src/evaluate.cu:709-711,747-748,794-795): the scale
    Q -> Ql is round(Ql x / Q) (within the floating-point carry), the expansion is times the dropped primes, the
    leveled multiply decrypts to the product, and the leveled key switch relinearises."""
    name, plain_t = "bfv13_50", 65537
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ql = size_q - drop
    q = [int(p) for p in primes[:size_q]]
    oc = oracle_ctx(name)
    hq = O.HpsOverQ(oc, plain_t, ql)
    assert hq.size_r == ql
    Q, Ql, D = 1, 1, 1
    for i, p in enumerate(q):
        Q *= p
        if i < ql:
            Ql *= p
        else:
            D *= p
    r = rng_for(83)
    x = uniform_poly(r, q, n)
    y = hq.scale_q_ql(x)
    z = hq.expand_ql_q(y)
    for k in range(0, n, 501):
        X, _ = crt_compose([x[l, k] for l in range(size_q)], q)
        Y, _ = crt_compose([y[l, k] for l in range(ql)], q[:ql])
        Xc = X - Q if X > Q // 2 else X                       # the conversion reads residues as the centred value
        want = (Xc * Ql * 2 + Q) // (2 * Q)                   # round(Ql * x / Q)
        dd = (Y - want) % Ql
        dd = dd - Ql if dd > Ql // 2 else dd
        assert abs(dd) <= 1
        Z, _ = crt_compose([z[l, k] for l in range(size_q)], q)
        assert Z == Y * D % Q
    # leveled multiply of genuine encryptions
    delta = Q // plain_t
    s_small = r.integers(-1, 2, n)
    sk_ntt = oc.nwt_forward(np.stack([(s_small % p).astype(np.uint64) for p in q]), size_q, 0)
    s2 = oc.multiply(sk_ntt, sk_ntt, size_q)

    def encrypt(m):
        a = uniform_poly(r, q, n)
        e = r.integers(-3, 4, n)
        dm = np.stack([np.array([(delta * int(v) + int(ev)) % p for v, ev in zip(m, e)], dtype=np.uint64) for p in q])
        a_s = oc.nwt_backward(oc.multiply(oc.nwt_forward(a, size_q, 0), sk_ntt, size_q), size_q)
        return np.stack([oc.sub(dm, a_s, size_q), a])

    big = int(O.get_primes(n, 60, 1)[0])
    bc = O.Ctx(log_n, [big], 0)
    m1, m2 = r.integers(0, plain_t, n), r.integers(0, plain_t, n)
    pm = bc.nwt_backward(bc.multiply(bc.nwt_forward(m1.astype(np.uint64).reshape(1, n), 1, 0),
                                     bc.nwt_forward(m2.astype(np.uint64).reshape(1, n), 1, 0), 1), 1)[0]

    def check(parts):
        d_ntt = [oc.nwt_forward(p, size_q, 0) for p in parts]
        phase = oc.add(d_ntt[0], oc.multiply(d_ntt[1], sk_ntt, size_q), size_q)
        if len(parts) == 3:
            phase = oc.add(phase, oc.multiply(d_ntt[2], s2, size_q), size_q)
        phase = oc.nwt_backward(phase, size_q)
        for k in range(0, n, 61):
            v, _ = crt_compose([phase[l, k] for l in range(size_q)], q)
            got = ((v * plain_t + Q // 2) // Q) % plain_t
            w = int(pm[k])
            w = w - big if w > big // 2 else w
            assert got == w % plain_t, k

    d = hq.multiply(encrypt(m1), encrypt(m2))
    assert not d[:, ql:].any()                                # expanded: multiples of the dropped primes
    check([d[0], d[1], d[2]])
    # leveled relinearisation with genuine keys for s^2
    evk = _small_keys(oc, primes, n, size_q, size_p, r, s2, oc.nwt_forward(
        np.stack([(s_small % int(p)).astype(np.uint64) for p in primes]), len(primes), 0))
    tool = O.Tool(oc, ql)
    relin = hq.keyswitch_leveled(tool, d[:2], d[2], [evk[i] for i in range(tool.beta)])
    check([relin[0], relin[1]])
    # the fused multiply + relinearise keeps c2 at level l: a different ciphertext with the same plaintext
    fused = hq.mul_relin_leveled(tool, encrypt(m1), encrypt(m2), [evk[i] for i in range(tool.beta)])
    check([fused[0], fused[1]])


def test_gemm_mod_against_python_ints():
    q = int(O.get_primes(4096, 50, 1)[0])
    r = rng_for(200)
    A = r.integers(0, q, (5, 9), dtype=np.uint64)
    B = r.integers(0, q, (9, 7), dtype=np.uint64)
    want = [[sum(int(A[i, l]) * int(B[l, j]) for l in range(9)) % q for j in range(7)] for i in range(5)]
    assert O.gemm_mod(q, A, B).tolist() == want
    ones = np.ones((4, 256), dtype=np.uint64), np.ones((256, 4), dtype=np.uint64)
    assert (O.gemm_mod(q, *ones) == 256).all() and (O.gemm_mod(q, *ones, quirk=True) == 256).all()  # the benchmark's input
    big = r.integers(q // 2, q, (2, 256), dtype=np.uint64), r.integers(q // 2, q, (256, 2), dtype=np.uint64)
    assert not np.array_equal(O.gemm_mod(q, *big), O.gemm_mod(q, *big, quirk=True))   # the dropped carries show


def test_polymath_ext_restatements_agree_with_the_c_oracle():
    """oracle/polymath_ext.py (Python integers) against the C restatement where the two overlap."""
    from oracle import polymath_ext as X
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    L = 4
    r = rng_for(77)
    a, b, d = (uniform_poly(r, primes[:L], n) for _ in range(3))
    assert np.array_equal(X.add_and_negate(a, b, primes[:L]), oc.negate(oc.add(a, b, L), L))
    assert np.array_equal(X.multiply_and_add_negate(a, b, d, primes[:L]), oc.negate(oc.multiply_and_add(a, b, d, L), L))
    assert np.array_equal(X.multiply_and_scale_add(a, b, d, 1, primes[:L]), oc.multiply_and_add(a, b, d, L))
    s = np.array([5] * L, dtype=np.uint64)
    assert np.array_equal(X.multiply_uniform_scalar(a, 5, primes[:L]), oc.multiply_scalar(a, s, L))
    assert np.array_equal(X.sub_and_scale(a, b, s, primes[:L]), oc.multiply_scalar(oc.sub(a, b, L), s, L))
    c1, c2 = np.stack([a, b]), np.stack([b, d])
    assert np.array_equal(X.tensor_prod_mxn(c1, c2, primes[:L]), oc.tensor_prod_2x2(c1, c2, L))
    assert np.array_equal(X.add_many([c1, c2, c1], 1, primes[:L]), oc.add(oc.add(b, d, L), b, L))


@pytest.mark.parametrize("t", [65537, 1 << 20, 786433])
def test_plaintext_ops_of_the_oracle(t):
    """BFV add_plain adds exactly ceil(Q m / t) (the scaling variant of src/scalingvariant.cu:10-60); the C restatements
    agree with the Python-integer formulas of oracle/polymath_ext.py; multiply_plain is the negacyclic product with the
    centred plaintext."""
    from oracle import polymath_ext as X
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    ql = 3
    r = rng_for(88)
    ct0 = uniform_poly(r, primes[:ql], n)
    m = r.integers(0, t, n, dtype=np.uint64)
    m[:3] = [0, 1, t - 1]
    Q = 1
    for q in primes[:ql]:
        Q *= int(q)
    neg = t - Q % t
    t_inv = [pow(t, -1, int(q)) for q in primes[:ql]]
    for sub in (False, True):
        got = oc.bfv_add_plain(ct0, m, t, subtract=sub)
        assert np.array_equal(got, X.bfv_timesQ_overt(ct0, m, neg, t_inv, t, primes[:ql], sub=sub))
    delta = oc.sub(oc.bfv_add_plain(ct0, m, t), ct0, ql)
    for k in (0, 1, 2, 5, 77, n - 1):
        v, _ = crt_compose([delta[l, k] for l in range(ql)], primes[:ql])
        assert v == (-(-Q * int(m[k]) // t)) % Q          # ceil(Q m / t)
    lifted = oc.bgv_lift_plain(m, ql)
    assert np.array_equal(lifted, oc.nwt_forward(np.stack([m % np.uint64(q) for q in primes[:ql]]), ql, 0))
    if t < min(primes[:ql]):
        ct = np.stack([ct0, uniform_poly(r, primes[:ql], n)])
        prod = oc.bfv_multiply_plain(ct, m, t)
        centred = X.abs_plain(m, (t + 1) >> 1, [int(q) - t for q in primes[:ql]])
        for p in range(2):
            want = oc.nwt_backward(oc.multiply(oc.nwt_forward(ct[p], ql, 0), oc.nwt_forward(centred, ql, 0), ql), ql, 0)
            assert np.array_equal(prod[p], want)


def test_every_literal_default_prime_of_the_reference_is_an_ntt_prime_for_the_oracle():
    """All three literal tables of src/host/globals.cu:51-361 (fixture: tests/golden/default_coeff_modulus.json, extracted
    by tests/golden/make_default_moduli.py): the oracle's primality test accepts every entry, every entry is = 1 mod 2N
    (so the oracle's root search and table builder apply to it), entries of one row are distinct, and the row respects
    the reference's own bit budget (include/host/hestdparms.h; one row of the reference is a bit over: recorded)."""
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "default_coeff_modulus.json")))
    rows = 0
    for level, table in doc["coeff_modulus"].items():
        for deg, primes in table.items():
            n, ps = int(deg), [int(p, 16) for p in primes]
            assert len(set(ps)) == len(ps)
            prod = 1
            for p in ps:
                assert O.is_prime(p), (level, deg, hex(p))
                assert (p - 1) % (2 * n) == 0, (level, deg, hex(p))
                assert p.bit_length() <= 60
                prod *= p
            assert prod.bit_length() <= doc["max_bit_count"][level][deg] or (level, deg) == ("192", "8192")
            rows += 1
    assert rows == 21
    # the rows BASELINE config 0 uses: tables and a round trip through the oracle's NTT for BFVDefault(4096) and (8192)
    for deg in ("4096", "8192"):
        ps = [int(p, 16) for p in doc["coeff_modulus"]["128"][deg]]
        n = int(deg)
        c = O.Ctx(n.bit_length() - 1, ps, 1)
        x = np.stack([np.random.default_rng(7).integers(0, q, n, dtype=np.uint64) for q in ps])
        assert np.array_equal(c.nwt_backward(c.nwt_forward(x, len(ps), 0), len(ps), 0), x)
        for i, q in enumerate(ps):
            assert pow(O.minimal_primitive_root(2 * n, q), n, q) == q - 1
