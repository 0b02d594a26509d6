"""pyPhantom (pybind11 over host/phantom.h): the reference's Python names for the hot path
(python/src/binding.cu:8-166 subset), checked against the oracle like python/examples/ckks.py's flow
(multiply_and_relin, rescale_to_next, rotate, hoisting) but with synthetic ciphertexts and keys."""
import numpy as np
import pytest

from oracle import oracle as O
from util import oracle_ctx, primes_of, rng_for, uniform_poly


def test_module_surface():
    from phantom_fhe_amd import pyPhantom as ph
    for name in ("scheme_type", "mul_tech_type", "modulus", "params", "context", "ciphertext", "relin_key",
                 "galois_key", "create_coeff_modulus", "get_elt_from_step", "get_elts_from_steps", "negate", "add",
                 "sub", "multiply", "multiply_and_relin", "relinearize", "rescale_to_next", "mod_switch_to_next",
                 "apply_galois", "rotate", "hoisting", "plaintext", "add_plain", "sub_plain", "multiply_plain", "add_many",
                 "mod_switch_to", "create_plain_modulus", "sec_level_type"):
        assert hasattr(ph, name), name
    assert ph.get_elt_from_step(1, 4096) == 5 and ph.get_elt_from_step(0, 4096) == 8191
    t = ph.create_plain_modulus(4096, 20).value()          # PlainModulus::Batching(4096, 20), examples/1_bfv.cu
    assert t % 8192 == 1 and t.bit_length() == 20
    mods = ph.create_coeff_modulus(1 << 14, [60] + [40] * 6 + [60])
    assert mods[0].value() == 1152921504606683137          # SURVEY.md 8(c) known answer


@pytest.mark.gpu
def test_ckks_flow(gpu):
    from phantom_fhe_amd import pyPhantom as ph
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n, size_q = 1 << log_n, len(primes) - size_p
    parms = ph.params(ph.scheme_type.ckks)
    parms.set_poly_modulus_degree(n)
    parms.set_special_modulus_size(size_p)
    parms.set_coeff_modulus(ph.create_coeff_modulus(n, [60, 40, 40, 40, 40, 40, 60, 60]))
    ctx = ph.context(parms)
    assert ctx.total_parm_size() == 1 + size_q and ctx.coeff_modulus_size(1) == size_q
    oc = oracle_ctx(name)
    tool = O.Tool(oc, size_q)
    r = rng_for(90)
    dnum = size_q // size_p

    def keys():
        return np.stack([np.stack([uniform_poly(r, primes, n), uniform_poly(r, primes, n)]) for _ in range(dnum)])

    evk, g1, g2 = keys(), keys(), keys()
    rlk = ph.relin_key(); rlk.load(ctx, evk)
    e1, e2 = ph.get_elt_from_step(1, n), ph.get_elt_from_step(2, n)
    glk = ph.galois_key(); glk.load(ctx, e1, g1); glk.load(ctx, e2, g2)
    h1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    h2 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    a, b = ph.ciphertext(), ph.ciphertext()
    a.load(ctx, 1, h1); b.load(ctx, 1, h2)
    a.set_scale(2.0 ** 40); b.set_scale(2.0 ** 40)
    assert np.array_equal(ph.add(ctx, a, b).to_numpy(), np.stack([oc.add(h1[i], h2[i], size_q) for i in range(2)]))
    prod = ph.multiply_and_relin(ctx, a, b, rlk)
    t3 = oc.tensor_prod_2x2(h1, h2, size_q)
    ref = tool.keyswitch_inplace(t3[:2], t3[2], [evk[i] for i in range(tool.beta)], O.CKKS)
    assert np.array_equal(prod.to_numpy(), ref) and prod.scale() == 2.0 ** 80
    res = ph.rescale_to_next(ctx, prod)
    assert np.array_equal(res.to_numpy(), tool.rescale_ntt(ref, 2)) and res.chain_index() == 2
    fused = ph.multiply_relin_rescale(ctx, a, b, rlk)           # extension: the same ciphertext from one fused call
    assert np.array_equal(fused.to_numpy(), res.to_numpy()) and fused.chain_index() == 2 and fused.scale() == res.scale()
    assert np.array_equal(ph.relinearize_rescale(ctx, ph.multiply(ctx, a, b), rlk).to_numpy(), res.to_numpy())
    rot = ph.rotate(ctx, a, 1, glk)
    tab = O.galois_ntt_table(log_n, e1)
    c0 = O.apply_galois_ntt(h1[0], tab, n, size_q)
    c1 = O.apply_galois_ntt(h1[1], tab, n, size_q)
    want = tool.keyswitch_inplace(np.stack([c0, np.zeros_like(c0)]), c1, [g1[i] for i in range(tool.beta)], O.CKKS)
    assert np.array_equal(rot.to_numpy(), want)
    hs = ph.hoisting(ctx, a, glk, [1, 2])
    assert np.array_equal(hs.to_numpy(), tool.hoisting(h1, [e1, e2], [[g1[i] for i in range(tool.beta)],
                                                                     [g2[i] for i in range(tool.beta)]], O.CKKS))
    with pytest.raises(ValueError):
        ph.relinearize(ctx, a, rlk)          # std::invalid_argument -> ValueError, as pybind does for the reference


@pytest.mark.gpu
def test_keygen_and_files(gpu, tmp_path):
    """relin_key.generate (caller-supplied randomness) vs the oracle, and the reference's on-disk formats."""
    from phantom_fhe_amd import pyPhantom as ph
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n, size_q = 1 << log_n, len(primes) - size_p
    parms = ph.params(ph.scheme_type.ckks)
    parms.set_poly_modulus_degree(n)
    parms.set_special_modulus_size(size_p)
    parms.set_coeff_modulus(ph.create_coeff_modulus(n, [60, 40, 40, 40, 40, 40, 60, 60]))
    ctx = ph.context(parms)
    oc = oracle_ctx(name)
    r = rng_for(91)
    dnum = size_q // size_p
    s_small = r.integers(-1, 2, n)
    sk = oc.nwt_forward(np.stack([(s_small % int(q)).astype(np.uint64) for q in primes]), len(primes), 0)
    s2 = oc.multiply(sk[:size_q], sk[:size_q], size_q)
    a = np.stack([uniform_poly(r, primes, n) for _ in range(dnum)])
    e_small = r.integers(-3, 4, (dnum, n))
    e = np.stack([np.stack([(e_small[d] % int(q)).astype(np.uint64) for q in primes]) for d in range(dnum)])
    ref = oc.gen_kswitch_key(sk, s2, a, np.stack([oc.nwt_forward(e[d], len(primes), 0) for d in range(dnum)]))
    rlk = ph.relin_key()
    rlk.generate(ctx, sk, s2, a, e)
    assert np.array_equal(rlk.to_numpy(), ref)
    path = str(tmp_path / "rlk.bin")
    rlk.save(path)
    import os
    header = 4 * 8 + 8 + 8 + 8 + 2
    assert os.path.getsize(path) == 8 + dnum * (header + 2 * len(primes) * n * 8)
    back = ph.relin_key()
    back.load_file(path)
    assert np.array_equal(back.to_numpy(), ref)
    ct = ph.ciphertext()
    h = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct.load(ctx, 1, h)
    ct.set_scale(2.0 ** 40)
    cpath = str(tmp_path / "ct.bin")
    ct.save(cpath)
    ct2 = ph.ciphertext()
    ct2.load_file(cpath)
    assert np.array_equal(ct2.to_numpy(), h) and ct2.scale() == 2.0 ** 40 and ct2.chain_index() == 1


@pytest.mark.gpu
def test_plaintext_flows(gpu):
    """add_plain / sub_plain / multiply_plain / add_many / mod_switch_to through the reference's Python names
    (python/examples/bfv.py, bgv.py, ckks.py use them), BGV and BFV with coefficient plaintexts modulo t."""
    from phantom_fhe_amd import pyPhantom as ph
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n, size_q = 1 << log_n, len(primes) - size_p
    oc = oracle_ctx(name)
    r = rng_for(91)
    t = 65537
    h1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    m = r.integers(0, t, (1, n), dtype=np.uint64)

    def make(scheme):
        parms = ph.params(scheme)
        parms.set_poly_modulus_degree(n)
        parms.set_special_modulus_size(size_p)
        parms.set_coeff_modulus(ph.create_coeff_modulus(n, [60, 40, 40, 40, 40, 40, 60, 60]))
        if scheme != ph.scheme_type.ckks:
            parms.set_plain_modulus(ph.modulus(t))
        return ph.context(parms)

    # BGV
    ctx = make(ph.scheme_type.bgv)
    a = ph.ciphertext(); a.load(ctx, 1, h1); a.set_correction_factor(7)
    pt = ph.plaintext(); pt.load(m)
    lifted = oc.bgv_lift_plain(m[0], size_q)
    scaled = oc.multiply_scalar(lifted, np.array([7] * size_q, dtype=np.uint64), size_q)
    got = ph.add_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got[0], oc.add(h1[0], scaled, size_q)) and np.array_equal(got[1], h1[1])
    got = ph.sub_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got[0], oc.sub(h1[0], scaled, size_q))
    got = ph.multiply_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got, np.stack([oc.multiply(h1[p], lifted, size_q) for p in range(2)]))
    # BFV
    ctx = make(ph.scheme_type.bfv)
    a = ph.ciphertext(); a.load(ctx, 1, h1); a.set_ntt_form(False)
    got = ph.add_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got[0], oc.bfv_add_plain(h1[0], m[0], t)) and np.array_equal(got[1], h1[1])
    got = ph.multiply_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got, oc.bfv_multiply_plain(h1, m[0], t))
    lower = ph.mod_switch_to(ctx, a, 3)
    assert lower.chain_index() == 3 and lower.coeff_modulus_size() == size_q - 2
    # CKKS
    ctx = make(ph.scheme_type.ckks)
    a = ph.ciphertext(); a.load(ctx, 1, h1); a.set_scale(2.0 ** 40)
    hp = uniform_poly(r, primes[:size_q], n)
    pt = ph.plaintext(); pt.load(hp, 1, 2.0 ** 40)
    got = ph.add_plain(ctx, a, pt).to_numpy()
    assert np.array_equal(got[0], oc.add(h1[0], hp, size_q))
    prod = ph.multiply_plain(ctx, a, pt)
    assert prod.scale() == 2.0 ** 80
    assert np.array_equal(prod.to_numpy(), np.stack([oc.multiply(h1[p], hp, size_q) for p in range(2)]))
    total = ph.add_many(ctx, [a, a, a])
    assert np.array_equal(total.to_numpy()[0], oc.add(oc.add(h1[0], h1[0], size_q), h1[0], size_q))
    low = ph.mod_switch_to(ctx, pt, 2)
    assert low.chain_index() == 2 and np.array_equal(low.to_numpy(), hp[: size_q - 1])
