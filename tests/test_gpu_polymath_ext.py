"""GPU parity for the rest of the reference's residue-wise kernels (src/polymath.cu) vs oracle/polymath_ext.py."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import polymath_ext as X
from util import primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of("hyb12_a2")          # 60/40-bit mix: the largest and smallest supported sizes
    ctx = P.PhantomContext(log_n, list(primes), size_p, device=gpu)
    return P, ctx, 1 << log_n, [int(q) for q in primes], gpu


def _edge(rng, primes, n):
    x = uniform_poly(rng, primes, n)
    x[:, :4] = 0
    x[:, 4:8] = np.array(primes, dtype=np.uint64)[:, None] - 1
    return x


def test_additive_family(env):
    P, ctx, n, primes, gpu = env
    L = 5
    r = rng_for(300)
    a, b, c = (_edge(r, primes[:L], n) for _ in range(3))
    dev = lambda v: P.to_device(np.ascontiguousarray(v), gpu)
    out = dev(np.zeros_like(a))
    ctx.add_and_negate_rns_poly(dev(a), dev(b), out, L)
    assert np.array_equal(P.to_host(out), X.add_and_negate(a, b, primes[:L]))
    ctx.add_and_negate_rns_poly(dev(a[2:4]), dev(b[2:4]), out[:2], 2, 2)            # a limb window
    assert np.array_equal(P.to_host(out)[:2], X.add_and_negate(a[2:4], b[2:4], primes[2:4]))
    c1, c2 = np.stack([a, b]), np.stack([b, c])
    out2 = dev(np.zeros_like(c1))
    ctx.add_std_cipher(dev(c1), dev(c2), out2, L)
    assert np.array_equal(P.to_host(out2), X.add_std_cipher(c1, c2, primes[:L]))
    many = [np.stack([_edge(r, primes[:L], n) for _ in range(2)]) for _ in range(7)]
    res = dev(np.zeros_like(many[0]))
    d_many = [dev(m) for m in many]
    for poly in range(2):
        ctx.add_many_rns_poly(d_many, res, poly, L)
    got = P.to_host(res)
    for poly in range(2):
        assert np.array_equal(got[poly], X.add_many(many, poly, primes[:L]))
    with pytest.raises(ValueError):
        ctx.add_many_rns_poly([], res, 0, L)


def test_multiplicative_family(env):
    P, ctx, n, primes, gpu = env
    L = 6
    r = rng_for(301)
    a, b, d = (_edge(r, primes[:L], n) for _ in range(3))
    dev = lambda v: P.to_device(np.ascontiguousarray(v), gpu)
    out = dev(np.zeros_like(a))
    for scale in (0, 1, 65537, (1 << 60) - 93, (1 << 64) - 1):
        ctx.multiply_uniform_scalar_rns_poly(dev(a), scale, out, L)
        assert np.array_equal(P.to_host(out), X.multiply_uniform_scalar(a, scale, primes[:L]))
        ctx.multiply_scalar_and_add_rns_poly(dev(a), dev(b), scale, out, L)
        assert np.array_equal(P.to_host(out), X.multiply_scalar_and_add(a, b, scale, primes[:L]))
        ctx.multiply_scalar_and_sub_rns_poly(dev(a), dev(b), scale, out, L)
        assert np.array_equal(P.to_host(out), X.multiply_scalar_and_sub(a, b, scale, primes[:L]))
        ctx.multiply_and_scale_add_rns_poly(dev(a), dev(b), dev(d), scale, out, L)
        assert np.array_equal(P.to_host(out), X.multiply_and_scale_add(a, b, d, scale, primes[:L]))
    ctx.multiply_and_add_negate_rns_poly(dev(a), dev(b), dev(d), out, L)
    assert np.array_equal(P.to_host(out), X.multiply_and_add_negate(a, b, d, primes[:L]))
    scale = np.array([int(r.integers(1, q)) for q in primes[:L]], dtype=np.uint64)
    shoup = np.array([O.compute_shoup(int(s), q) for s, q in zip(scale, primes[:L])], dtype=np.uint64)
    ctx.sub_and_scale_rns_poly(dev(a), dev(b), dev(scale), dev(shoup), out, L)
    assert np.array_equal(P.to_host(out), X.sub_and_scale(a, b, scale, primes[:L]))
    q3 = primes[3]
    ctx.sub_and_scale_single_mod_poly(dev(a[3]), dev(b[3]), int(scale[3]), int(shoup[3]), q3, out[0])
    assert np.array_equal(P.to_host(out)[0], X.sub_and_scale(a[3:4], b[3:4], scale[3:4], [q3])[0])


def test_plaintext_helpers(env):
    P, ctx, n, primes, gpu = env
    L = 4
    t = 65537
    r = rng_for(302)
    dev = lambda v: P.to_device(np.ascontiguousarray(v), gpu)
    ct = _edge(r, primes[:L], n)
    pt = r.integers(0, t, n, dtype=np.uint64)
    pt[:3] = [0, t - 1, 1]
    Ql = 1
    for q in primes[:L]:
        Ql *= q
    neg = (-Ql) % t
    t_inv = np.array([pow(t, -1, q) for q in primes[:L]], dtype=np.uint64)
    t_inv_s = np.array([O.compute_shoup(int(v), q) for v, q in zip(t_inv, primes[:L])], dtype=np.uint64)
    for sub in (False, True):
        d_ct = dev(ct)
        ctx.bfv_add_timesQ_overt(d_ct, dev(pt), neg, O.compute_shoup(neg, t), dev(t_inv), dev(t_inv_s), t, L, sub=sub)
        assert np.array_equal(P.to_host(d_ct), X.bfv_timesQ_overt(ct, pt, neg, t_inv, t, primes[:L], sub=sub))
    inc = np.array([q - t for q in primes[:L]], dtype=np.uint64)
    out = dev(np.zeros((L, n), dtype=np.uint64))
    ctx.abs_plain_rns_poly(dev(pt), (t + 1) >> 1, dev(inc), out, L)
    assert np.array_equal(P.to_host(out), X.abs_plain(pt, (t + 1) >> 1, inc))
    # BEHZ FastBconvSK fix-up with a 61-bit m_sk
    m_sk = int(O.get_primes(n, 61, 1)[0])
    alpha = r.integers(0, m_sk, n, dtype=np.uint64)
    alpha[:4] = [0, m_sk - 1, m_sk >> 1, (m_sk >> 1) + 1]
    pb = np.array([int(r.integers(0, q)) for q in primes[:L]], dtype=np.uint64)
    ctx.multiply_and_negated_add_rns_poly(dev(alpha), m_sk, dev(pb), dev(ct), out, L)
    assert np.array_equal(P.to_host(out), X.multiply_and_negated_add(alpha, m_sk, pb, ct, primes[:L]))


@pytest.mark.parametrize("m,k", [(2, 2), (3, 2), (2, 3), (1, 4), (4, 4), (8, 8)])
def test_tensor_prod_mxn(env, m, k):
    P, ctx, n, primes, gpu = env
    L = 3
    r = rng_for(310 + m * 10 + k)
    op1 = np.stack([_edge(r, primes[:L], n) for _ in range(m)])
    op2 = np.stack([_edge(r, primes[:L], n) for _ in range(k)])
    out = P.to_device(np.zeros((m + k - 1, L, n), dtype=np.uint64), gpu)
    ctx.tensor_prod_mxn_rns_poly(P.to_device(op1, gpu), m, P.to_device(op2, gpu), k, out, L)
    assert np.array_equal(P.to_host(out), X.tensor_prod_mxn(op1, op2, primes[:L]))
    if (m, k) == (2, 2):     # agrees with the 2x2 kernel of the hot path
        ref = P.to_device(np.zeros((3, L, n), dtype=np.uint64), gpu)
        ctx.tensor_prod_2x2_rns_poly(P.to_device(op1, gpu), P.to_device(op2, gpu), ref, L)
        assert np.array_equal(P.to_host(out), P.to_host(ref))
    if (m, k) == (8, 8):
        with pytest.raises(ValueError):
            ctx.tensor_prod_mxn_rns_poly(P.to_device(op1, gpu), 9, P.to_device(op2, gpu), k, out, L)


@pytest.mark.parametrize("name,ql,t", [("hyb12_a2", 6, 65537), ("hyb12_a2", 3, 1 << 20), ("c4_bfv15", 30, 1032193)])
def test_plaintext_ops(name, ql, t, gpu):
    """pha_bfv_add_plain / pha_bfv_multiply_plain / pha_bgv_lift_plain (src/scalingvariant.cu:10-60,
    src/evaluate.cu:1256-1300, 1150-1154) vs the C oracle."""
    import phantom_fhe_amd as P
    from util import oracle_ctx
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    ctx = P.PhantomContext(log_n, list(primes), size_p, device=gpu)
    r = rng_for(330 + ql)
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    m = r.integers(0, t, n, dtype=np.uint64)
    m[:3] = [0, 1, t - 1]
    d_m = P.to_device(m, gpu)
    with pytest.raises(ValueError):
        ctx.bfv_add_plain(ql, P.to_device(ct[0], gpu), d_m)          # no plain modulus yet
    ctx.set_plain_modulus(t)
    for sub in (False, True):
        d = P.to_device(ct[0], gpu)
        ctx.bfv_add_plain(ql, d, d_m, subtract=sub)
        assert np.array_equal(P.to_host(d), oc.bfv_add_plain(ct[0], m, t, subtract=sub))
    out = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.bgv_lift_plain(ql, d_m, out)
    assert np.array_equal(P.to_host(out), oc.bgv_lift_plain(m, ql))
    assert np.array_equal(P.to_host(d_m), m)
    d = P.to_device(ct, gpu)
    ctx.bfv_multiply_plain(ql, d, 2, d_m)
    assert np.array_equal(P.to_host(d), oc.bfv_multiply_plain(ct, m, t))
