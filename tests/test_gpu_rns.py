"""GPU parity: dyadic kernels, fast base conversion, mod-up, key-switch inner product, mod-down,
keyswitch_inplace, CKKS rescale and Galois permutations vs the CPU oracle, bit-exact, at the
BASELINE.json configurations (full size for C3/C4)."""
import numpy as np
import pytest

from oracle import oracle as O
from util import crt_compose, oracle_ctx, primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _ctx(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    return P.PhantomContext(log_n, list(primes), size_p, device=gpu)


@pytest.mark.parametrize("name", ["c1_bfv4096", "c3_ckks16"])
def test_dyadic(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    L = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(10)
    a, b, d = (uniform_poly(r, primes[:L], n) for _ in range(3))
    da, db, dd = (P.to_device(v, gpu) for v in (a, b, d))
    out = P.to_device(np.zeros_like(a), gpu)
    ctx.add_rns_poly(da, db, out, L); assert np.array_equal(P.to_host(out), oc.add(a, b, L))
    ctx.sub_rns_poly(da, db, out, L); assert np.array_equal(P.to_host(out), oc.sub(a, b, L))
    ctx.negate_rns_poly(da, out, L); assert np.array_equal(P.to_host(out), oc.negate(a, L))
    ctx.multiply_rns_poly(da, db, out, L); assert np.array_equal(P.to_host(out), oc.multiply(a, b, L))
    ctx.multiply_and_add_rns_poly(da, db, dd, out, L)
    assert np.array_equal(P.to_host(out), oc.multiply_and_add(a, b, d, L))
    sc = np.array([r.integers(0, int(q)) for q in primes[:L]], dtype=np.uint64)
    scs = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(sc, primes[:L])], dtype=np.uint64)
    ctx.multiply_scalar_rns_poly(da, P.to_device(sc, gpu), P.to_device(scs, gpu), out, L)
    assert np.array_equal(P.to_host(out), oc.multiply_scalar(a, sc, L))
    # edge values: 0 and q-1 everywhere
    z = np.zeros_like(a)
    m1 = np.stack([np.full(n, int(q) - 1, dtype=np.uint64) for q in primes[:L]])
    ctx.negate_rns_poly(P.to_device(z, gpu), out, L); assert np.array_equal(P.to_host(out), z)
    dm = P.to_device(m1, gpu)
    ctx.add_rns_poly(dm, dm, out, L); assert np.array_equal(P.to_host(out), oc.add(m1, m1, L))
    ctx.multiply_rns_poly(dm, dm, out, L); assert np.array_equal(P.to_host(out), oc.multiply(m1, m1, L))
    # tensor product, in place on operand1 as evaluate.cu:377-383 calls it
    ct1 = np.stack([a, b]); ct2 = np.stack([d, a])
    buf = P.to_device(np.concatenate([ct1, np.zeros((1, L, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(ct2, gpu), buf, L)
    assert np.array_equal(P.to_host(buf), oc.tensor_prod_2x2(ct1, ct2, L))
    buf = P.to_device(np.concatenate([ct1, np.zeros((1, L, n), dtype=np.uint64)]), gpu)
    ctx.tensor_square_2x2_rns_poly(buf, buf, L)
    assert np.array_equal(P.to_host(buf), oc.tensor_square_2x2(ct1, L))
    mm = np.stack([m1, m1])
    buf = P.to_device(np.concatenate([mm, np.zeros((1, L, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(mm, gpu), buf, L)   # unreduced (c0+c1) at its maximum
    assert np.array_equal(P.to_host(buf), oc.tensor_prod_2x2(mm, mm, L))
    buf = P.to_device(np.concatenate([mm, np.zeros((1, L, n), dtype=np.uint64)]), gpu)
    ctx.tensor_square_2x2_rns_poly(buf, buf, L)                        # 2 c0 c1 at its maximum
    assert np.array_equal(P.to_host(buf), oc.tensor_square_2x2(mm, L))
    # mixed extremes: the FP64 form's quotient estimate at both ends of its range (limbs below 2^50)
    ex = np.stack([np.stack([np.where(np.arange(n) % 3 == 0, 0, np.where(np.arange(n) % 3 == 1, 1, int(q) - 1)).astype(np.uint64)
                             for q in primes[:L]]) for _ in range(2)])
    ex[1] = ex[1][:, ::-1]
    buf = P.to_device(np.concatenate([ex, np.zeros((1, L, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(mm, gpu), buf, L)
    assert np.array_equal(P.to_host(buf), oc.tensor_prod_2x2(ex, mm, L))


def _keys(oc, rng, primes, n, size_q, size_p):
    """Uniform synthetic evaluation keys [dnum][2][QP][N] (arithmetic is data-independent)."""
    dnum = -(-size_q // size_p)     # (a special base wider than Q still needs one digit)
    return np.stack([np.stack([uniform_poly(rng, primes, n), uniform_poly(rng, primes, n)]) for _ in range(dnum)])


# (config, scheme, levels to test as number of live data limbs)
KS_CASES = [
    ("hyb12_a2", O.CKKS, [6, 5, 1]),
    ("hyb12_a2", O.BFV, [6]),
    ("hyb13_a3", O.CKKS, [9, 7, 4]),     # 7 and 4: short last digit (rns.cu:152-157)
    ("c1_bfv4096", O.BFV, [2]),          # alpha = 1 fast paths
    ("c1_bfv4096", O.CKKS, [2, 1]),
    ("c4_bfv15", O.BFV, [30]),
    ("c3_ckks16", O.CKKS, [45, 31]),
    ("hyb17_a2", O.CKKS, [4, 3]),        # N = 2^17 (3: a short last digit)
    ("hyb14_a2", O.CKKS, [8, 7]),        # beta = 4 under the fused mod-up + inner product
    ("hyb12_a2", O.BGV, [6, 5, 1]),      # t-corrected mod-down (bgv_moddown_kernel rns_bconv.cu:636-652)
    ("hyb13_a3", O.BGV, [9, 7]),
    ("c1_bfv4096", O.BGV, [2]),
    ("wide_p33", O.BFV, [6, 4]),         # alpha = 33 > 32: the generic converter, mod-down through the element-wise kernel
    ("wide_p33", O.CKKS, [6]),
    ("wide_p20", O.CKKS, [24, 21]),      # 17..32 special primes: split_kind 2 with a 32-entry row pitch next to a short last digit
    ("wide_p20", O.BFV, [24]),           # (+ the fused BFV mod-down epilogue on that kernel)
    ("wide_p20", O.BGV, [20]),
    ("p61_a2", O.CKKS, [6, 3]),          # 61-bit special primes: split_kind 2 (mod-up) and 3 (mod-down)
    ("p61_a2", O.BFV, [6]),
]
BGV_T = 65537


@pytest.mark.parametrize("name,scheme,levels", KS_CASES)
def test_keyswitch_stages(name, scheme, levels, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_qp, size_q = len(primes), len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(20)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    if scheme == O.BGV:
        ctx.set_plain_modulus(BGV_T)
    for ql in levels:
        tool = O.Tool(oc, ql)
        if scheme == O.BGV:
            tool.set_plain_modulus(BGV_T)
        qlp = ql + size_p
        beta = tool.beta
        assert ctx.beta(ql) == beta
        c2 = uniform_poly(r, primes[:ql], n)
        # mod-up (rns_bconv.cu:530-627)
        d_mu = P.to_device(np.zeros((beta, qlp, n), dtype=np.uint64), gpu)
        ctx.modup(ql, d_mu, P.to_device(c2, gpu), scheme)
        ref_mu = tool.modup(c2, scheme)
        assert np.array_equal(P.to_host(d_mu), ref_mu), f"modup ql={ql}"
        # inner product (eval_key_switch.cu:14-92)
        d_cx = P.to_device(np.zeros((2, qlp, n), dtype=np.uint64), gpu)
        ctx.key_switch_inner_prod(ql, d_cx, d_mu, rlk.public_keys_ptr)
        ref_cx = tool.key_switch_inner_prod(ref_mu, [evk[i] for i in range(beta)])
        assert np.array_equal(P.to_host(d_cx), ref_cx), f"inner product ql={ql}"
        # mod-down (rns_bconv.cu:776-828), in place like keyswitch_inplace does
        for i in range(2):
            ctx.moddown_from_NTT(ql, d_cx[i], d_cx[i], scheme)
            ref = tool.moddown_from_ntt(ref_cx[i], scheme)
            assert np.array_equal(P.to_host(d_cx[i])[:ql], ref), f"moddown ql={ql} poly={i}"
        # whole key switch on raw buffers (eval_key_switch.cu:95-182)
        ct = np.stack([uniform_poly(r, primes[:ql], n), uniform_poly(r, primes[:ql], n)])
        d_ct = P.to_device(ct, gpu)
        ctx.keyswitch_inplace(ql, d_ct, P.to_device(c2, gpu), rlk.public_keys_ptr, scheme)
        assert np.array_equal(P.to_host(d_ct), tool.keyswitch_inplace(ct, c2, [evk[i] for i in range(beta)], scheme))


@pytest.mark.parametrize("name,ql", [("hyb12_a2", 6), ("c3_ckks16", 45)])
def test_bconv_P_to_Ql(name, ql, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ctx = _ctx(name, gpu)
    src = uniform_poly(rng_for(30), primes[size_q:], n)
    dst = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.bconv_P_to_Ql(ql, dst, P.to_device(src, gpu))
    assert np.array_equal(P.to_host(dst), O.bconv(primes[size_q:], primes[:ql], src, n))


@pytest.mark.parametrize("name,ql", [("hyb12_a2", 6), ("hyb12_a2", 2), ("c3_ckks16", 45)])
def test_rescale(name, ql, gpu):
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(40)
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
    ctx.divide_and_round_q_last_ntt(ql, P.to_device(ct, gpu), 2, dst)
    assert np.array_equal(P.to_host(dst), tool.rescale_ntt(ct, 2))
    ctx.divide_and_round_q_last(ql, P.to_device(ct, gpu), 2, dst)      # BFV coefficient-domain switch
    assert np.array_equal(P.to_host(dst), tool.divide_and_round_q_last(ct, 2))


@pytest.mark.parametrize("name,ql,plain_t", [("hyb12_a2", 6, 65537), ("hyb12_a2", 2, 1 << 20), ("c4_bfv15", 30, 786433)])
def test_bgv_mod_switch(name, ql, plain_t, gpu):
    """mod_t_and_divide_q_last_ntt (src/rns.cu:1186-1236)."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(45)
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(3)])
    dst = P.to_device(np.zeros((3, ql - 1, n), dtype=np.uint64), gpu)
    with pytest.raises(ValueError):                       # no plain modulus yet
        ctx.mod_t_and_divide_q_last_ntt(ql, P.to_device(ct, gpu), 3, dst)
    ctx.set_plain_modulus(plain_t)
    tool = O.Tool(oc, ql).set_plain_modulus(plain_t)
    ctx.mod_t_and_divide_q_last_ntt(ql, P.to_device(ct, gpu), 3, dst)
    assert np.array_equal(P.to_host(dst), tool.mod_t_divide_q_last_ntt(ct, 3))
    with pytest.raises((ArithmeticError, ValueError)):    # shares a factor with the chain: logic_error
        ctx.set_plain_modulus(int(primes[1]))


def test_galois(gpu):
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = 6
    x = uniform_poly(rng_for(50), primes[:L], n)
    dx = P.to_device(x, gpu)
    out = P.to_device(np.zeros_like(x), gpu)
    for elt in (3, 5, 25, 2 * n - 1, 1):
        ctx.apply_galois_ntt(dx, out, elt, L)
        assert np.array_equal(P.to_host(out), O.apply_galois_ntt(x, O.galois_ntt_table(log_n, elt), n, L))
        ctx.apply_galois(dx, out, elt, L)
        assert np.array_equal(P.to_host(out), oc.apply_galois_coeff(x, elt, L))
    with pytest.raises(ValueError):
        ctx.apply_galois_ntt(dx, out, 4, L)


@pytest.mark.parametrize("name,scheme,ql,elts", [("hyb12_a2", O.CKKS, 6, [5, 25, 125]), ("hyb12_a2", O.CKKS, 3, [5]),
                                                 ("hyb12_a2", O.BFV, 6, [5, 8191]), ("c1_bfv4096", O.CKKS, 2, [5, 25]),
                                                 ("hyb12_a2", O.BGV, 6, [5, 25]),
                                                 ("c3_ckks16", O.CKKS, 45, [5, 25, 125, 625])])
def test_hoisting(name, scheme, ql, elts, gpu):
    """hoisting_inplace (src/evaluate.cu:1670-1866): ct <- sum_e rotate_e(ct) with one shared mod-up."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        ctx.set_plain_modulus(BGV_T)
        tool.set_plain_modulus(BGV_T)
    r = rng_for(70)
    glk = [_keys(oc, r, primes, n, size_q, size_p) for _ in elts]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_ct = P.to_device(ct, gpu)
    keys = [P.PhantomRelinKey.from_numpy(k, gpu) for k in glk]
    ctx.hoisting(ql, d_ct, elts, keys, scheme)
    ref = tool.hoisting(ct, elts, [[k[i] for i in range(tool.beta)] for k in glk], scheme)
    assert np.array_equal(P.to_host(d_ct), ref)


@pytest.mark.parametrize("name,scheme,ql,n_rot,ident", [
    ("hyb12_a2", O.CKKS, 6, 3, True), ("hyb12_a2", O.CKKS, 3, 2, False), ("hyb12_a2", O.BGV, 5, 3, True),
    ("hyb13_a3", O.CKKS, 7, 4, True), ("hyb12_a2", O.CKKS, 2, 0, True),
    ("hyb12_a2", O.CKKS, 4, 127, True),          # config 5 shape: 128 diagonals (the kernel folds every 63 / 48 terms)
    ("c3_ckks16", O.CKKS, 45, 2, True)])
def test_hoisting_weighted(name, scheme, ql, n_rot, ident, gpu):
    """pha_hoisting_weighted (build-defined, BASELINE config 5): sum_e w_e (.) rotate_e(ct) vs the oracle's
    composition; uniform random weights over [Q_l || P] (the arithmetic does not care that plaintexts are small)."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        ctx.set_plain_modulus(BGV_T)
        tool.set_plain_modulus(BGV_T)
    r = rng_for(170 + n_rot)
    rot = [int(pow(5, i + 1, 2 * n)) for i in range(n_rot)]
    elts = ([1] if ident else []) + rot
    if n_rot > 8:                                    # many diagonals: a few distinct keys, reused
        pool = [_keys(oc, r, primes, n, size_q, size_p) for _ in range(3)]
        glk_rot = [pool[i % 3] for i in range(n_rot)]
        pool_w = [uniform_poly(r, [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]], n) for _ in range(5)]
        weights = [pool_w[i % 5] for i in range(len(elts))]
        dev_pool = [P.PhantomRelinKey.from_numpy(k, gpu) for k in pool]
        dev_keys_rot = [dev_pool[i % 3] for i in range(n_rot)]
        dev_pool_w = [P.to_device(w, gpu) for w in pool_w]
        dev_w = [dev_pool_w[i % 5] for i in range(len(elts))]
    else:
        glk_rot = [_keys(oc, r, primes, n, size_q, size_p) for _ in rot]
        qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
        weights = [uniform_poly(r, qlp_primes, n) for _ in elts]
        dev_keys_rot = [P.PhantomRelinKey.from_numpy(k, gpu) for k in glk_rot]
        dev_w = [P.to_device(w, gpu) for w in weights]
    glk = ([None] if ident else []) + [[k[i] for i in range(tool.beta)] for k in glk_rot]
    dev_keys = ([None] if ident else []) + dev_keys_rot
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_ct = P.to_device(ct, gpu)
    ctx.hoisting_weighted(ql, d_ct, elts, dev_keys, dev_w, scheme)
    ref = tool.hoisting_weighted(ct, elts, glk, weights, scheme)
    assert np.array_equal(P.to_host(d_ct), ref)
    with pytest.raises(ValueError):
        ctx.hoisting_weighted(ql, d_ct, elts, dev_keys, dev_w, O.BFV)       # NTT-form schemes only
    if n_rot:
        with pytest.raises(ArithmeticError):         # std::logic_error, like hoisting_inplace (evaluate.cu:1783)
            ctx.hoisting_weighted(ql, d_ct, [rot[0]], [None], dev_w[:1], scheme)  # a rotation without its key


def test_hommul_relin_rescale_c3(gpu):
    """CKKS HomMul + relinearize + rescale at N=2^16, 45 limbs, stage by stage."""
    import phantom_fhe_amd as P
    name = "c3_ckks16"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ql = size_q
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(60)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    ct1 = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    buf = P.to_device(np.concatenate([ct1, np.zeros((1, ql, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(ct2, gpu), buf, ql)                  # multiply_inplace
    ref3 = oc.tensor_prod_2x2(ct1, ct2, ql)
    ctx.keyswitch_inplace(ql, buf, buf[2], rlk.public_keys_ptr, O.CKKS)              # relinearize_inplace
    ref2 = tool.keyswitch_inplace(ref3[:2], ref3[2], [evk[i] for i in range(tool.beta)], O.CKKS)
    assert np.array_equal(P.to_host(buf)[:2], ref2)
    dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
    ctx.divide_and_round_q_last_ntt(ql, buf[:2].contiguous(), 2, dst)                 # rescale_to_next
    assert np.array_equal(P.to_host(dst), tool.rescale_ntt(ref2, 2))


# (config, live data limbs, batch): alpha = 2 / 3 / 15 incl. short last digits, the alpha = 1 fallback, both BASELINE sizes
KSR_CASES = [("hyb12_a2", 6, 1), ("hyb12_a2", 5, 2), ("hyb12_a2", 2, 1), ("hyb13_a3", 9, 1), ("hyb13_a3", 7, 3), ("hyb13_a3", 4, 1),
             ("c1_bfv4096", 2, 2), ("c2_ckks14", 8, 2), ("hyb14_a4", 8, 1), ("hyb14_a4", 5, 2), ("c4_bfv15", 30, 2), ("c4_bfv15", 17, 1), ("c3_ckks16", 45, 1), ("c3_ckks16", 31, 2),
             ("c3_ckks16", 45, 4), ("hyb17_a2", 4, 1), ("hyb17_a2", 3, 2), ("hyb14_a2", 8, 1),   # 2 x 4 = 8 polynomials at N = 2^16: the twiddle-resident batched contiguous pass (r04) with the key switch + rescale epilogue
             # r05: batches whose mod-up launches >= 1024 workgroups take modup_conv_s1_kernel (base conversion fused with the forward
             # transform's strided pass): 6 x 3 digits at the top level; 8 x 3 digits at level 31, whose last digit is ONE limb
             # r06: from 8 ciphertexts on (16 polynomials = 1024 workgroups) the rescale's / mod-down's P -> Ql conversion is the load of the
             # final transform's strided pass too (the kernel's FORM 1 / 2), and the inner product reads the digits' own limbs from c2:
             # the top level runs the 15-input instantiations of all three forms in the PRODUCT library
             ("c3_ckks16", 45, 8), ("c3_ckks16", 31, 8),
             ("hyb16_a12", 24, 8)]    # alpha = 12: the 16-input instantiation, 8 x 2 digits = 1024 workgroups in the product library


@pytest.mark.parametrize("name,ql,batch", KSR_CASES)
def test_keyswitch_rescale_equals_keyswitch_then_rescale(name, ql, batch, gpu):
    """pha_keyswitch_rescale(_batched) -- ONE forward transform for mod-down + rescale -- against the oracle's composition
    keyswitch_inplace (eval_key_switch.cu:95-182) then divide_and_round_q_last_ntt (rns.cu:1160-1184), bit for bit; the inputs
    are left as they were; the two-call path of the library gives the same words."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(170 + ql)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    ct = np.stack([np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)]) for _ in range(batch)])
    c2 = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(batch)])
    if name == "hyb12_a2" and ql == 6:       # extreme residues: everything at q - 1
        ct[0] = np.stack([np.stack([np.full(n, int(q) - 1, dtype=np.uint64) for q in primes[:ql]])] * 2)
        c2[0] = ct[0][0]
    d_ct, d_c2 = P.to_device(ct, gpu), P.to_device(c2, gpu)
    dst = P.to_device(np.zeros((batch, 2, ql - 1, n), dtype=np.uint64), gpu)
    if batch == 1:
        ctx.keyswitch_rescale(ql, d_ct[0], d_c2[0], rlk.public_keys_ptr, dst[0])
    else:
        ctx.keyswitch_rescale_batched(ql, d_ct, d_c2, batch, rlk.public_keys_ptr, dst)
    got = P.to_host(dst)
    keys = [evk[i] for i in range(tool.beta)]
    for b in range(batch):
        ref = tool.rescale_ntt(tool.keyswitch_inplace(ct[b], c2[b], keys, O.CKKS), 2)
        assert np.array_equal(got[b], ref), f"ciphertext {b}"
    assert np.array_equal(P.to_host(d_ct), ct) and np.array_equal(P.to_host(d_c2), c2)      # inputs untouched
    two = P.to_device(np.zeros((batch, 2, ql - 1, n), dtype=np.uint64), gpu)
    ctx.keyswitch_inplace_batched(ql, d_ct, d_c2, batch, rlk.public_keys_ptr, O.CKKS)
    ctx.divide_and_round_q_last_ntt(ql, d_ct, 2 * batch, two)
    assert np.array_equal(P.to_host(two), got)
    with pytest.raises(ValueError):           # dst must not overlap ct
        ctx.keyswitch_rescale(ql, d_ct[0], d_c2[0], rlk.public_keys_ptr, d_ct[0])


def test_keyswitch_rescale_replays_from_a_hip_graph(gpu):
    """pha_keyswitch_rescale enqueues plain kernel launches on the caller's stream (arguments and work maps by value, scratch from the
    stream's arena, no host synchronisation): captured into a hipGraph after one warm-up call on that stream, it replays on NEW inputs
    to what the eager call gives -- the fused mod-up + inner product with the folded inverse pass included (N = 2^14, alpha 4)."""
    import torch
    import phantom_fhe_amd as P
    name, ql = "hyb14_a4", 8
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(733)
    rlk = P.PhantomRelinKey.from_numpy(_keys(oc, r, primes, n, size_q, size_p), gpu)
    ins = [(np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)]), uniform_poly(r, primes[:ql], n)) for _ in range(3)]
    want = []
    for ct, c2 in ins:
        dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
        ctx.keyswitch_rescale(ql, P.to_device(ct, gpu), P.to_device(c2, gpu), rlk.public_keys_ptr, dst)
        want.append(P.to_host(dst))
    d_ct, d_c2 = P.to_device(ins[0][0], gpu), P.to_device(ins[0][1], gpu)
    out = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):
        ctx.keyswitch_rescale(ql, d_ct, d_c2, rlk.public_keys_ptr, out)      # warm-up: the stream's scratch arena exists before the capture
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ctx.keyswitch_rescale(ql, d_ct, d_c2, rlk.public_keys_ptr, out)
    for i in (1, 2, 0):
        d_ct.copy_(P.to_device(ins[i][0], gpu))
        d_c2.copy_(P.to_device(ins[i][1], gpu))
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(P.to_host(out), want[i]), i


def _ternary_sk(oc, rng, primes, n):
    s_small = rng.integers(-1, 2, n)
    sk = np.stack([(s_small % int(q)).astype(np.uint64) for q in primes])
    return oc.nwt_forward(sk, len(primes), 0)


def _noise(rng, primes, n, count):
    e_small = rng.integers(-3, 4, (count, n))
    return np.stack([np.stack([(e_small[d] % int(q)).astype(np.uint64) for q in primes]) for d in range(count)])


@pytest.mark.parametrize("name,scheme", [("hyb12_a2", O.CKKS), ("hyb12_a2", O.BGV), ("hyb13_a3", O.CKKS),
                                         ("c1_bfv4096", O.BFV)])
def test_generate_one_kswitch_key(name, scheme, gpu):
    """generate_one_kswitch_key (src/secretkey.cu:297-341) with caller-supplied randomness vs the oracle."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    dnum = size_q // size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(80)
    sk_ntt = _ternary_sk(oc, r, primes, n)
    new_key = oc.multiply(sk_ntt[:size_q], sk_ntt[:size_q], size_q)
    a = np.stack([uniform_poly(r, primes, n) for _ in range(dnum)])
    e = _noise(r, primes, n, dnum)
    e_ref = e
    if scheme == O.BGV:
        ctx.set_plain_modulus(BGV_T)
        e_ref = np.stack([np.stack([(e[d, j].astype(object) * BGV_T % int(q)).astype(np.uint64)
                                    for j, q in enumerate(primes)]) for d in range(dnum)])
    e_ntt = np.stack([oc.nwt_forward(e_ref[d], len(primes), 0) for d in range(dnum)])
    ref = oc.gen_kswitch_key(sk_ntt, new_key, a, e_ntt)
    rlk = ctx.generate_one_kswitch_key(P.to_device(sk_ntt, gpu), P.to_device(new_key, gpu), P.to_device(a, gpu),
                                       P.to_device(e, gpu), scheme)
    for d in range(dnum):
        assert np.array_equal(P.to_host(rlk.public_keys[d]), ref[d]), f"digit {d}"


def test_c3_hommul_decrypts_with_generated_keys(gpu):
    """SURVEY 8(0) C3 criterion: with keys generated by the build itself, HomMul -> relinearize -> rescale
    of two genuine encryptions decrypts to m1*m2/q_last (negacyclic product) up to noise, at N=2^16, 45 limbs."""
    import phantom_fhe_amd as P
    from util import crt_compose
    name = "c3_ckks16"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ql, dnum = size_q, size_q // size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(90)
    sk_ntt = _ternary_sk(oc, r, primes, n)
    d_sk = P.to_device(sk_ntt, gpu)
    d_s2 = P.to_device(np.zeros((size_q, n), dtype=np.uint64), gpu)
    ctx.multiply_rns_poly(d_sk, d_sk, d_s2, size_q)
    a = np.stack([uniform_poly(r, primes, n) for _ in range(dnum)])
    rlk = ctx.generate_one_kswitch_key(d_sk, d_s2, P.to_device(a, gpu), P.to_device(_noise(r, primes, n, dnum), gpu),
                                       O.CKKS)
    q = primes[:ql]

    def encrypt(m_small):
        m = np.stack([(m_small % int(p)).astype(np.uint64) for p in q])
        c1 = uniform_poly(r, q, n)
        e = _noise(r, q, n, 1)[0]
        me_ntt = oc.nwt_forward(oc.add(m, e, ql), ql, 0)
        c0 = oc.sub(me_ntt, oc.multiply(c1, sk_ntt[:ql], ql), ql)       # c0 = m + e - c1*s
        return np.stack([c0, c1]), oc.nwt_forward(m, ql, 0)

    m1 = r.integers(-(1 << 30), 1 << 30, n)
    m2 = r.integers(-(1 << 30), 1 << 30, n)
    ct1, m1_ntt = encrypt(m1)
    ct2, m2_ntt = encrypt(m2)
    buf = P.to_device(np.concatenate([ct1, np.zeros((1, ql, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(ct2, gpu), buf, ql)
    ctx.keyswitch_inplace(ql, buf[:2], buf[2], rlk.public_keys_ptr, O.CKKS)
    out = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
    ctx.divide_and_round_q_last_ntt(ql, buf[:2].contiguous(), 2, out)
    res = P.to_host(out)
    nl = ql - 1
    phase = oc.nwt_backward(oc.add(res[0], oc.multiply(res[1], sk_ntt[:nl], nl), nl), nl)
    prod = oc.nwt_backward(oc.multiply(m1_ntt, m2_ntt, ql), ql)          # exact m1*m2 mod (X^N+1), in RNS
    Q, Qn = 1, 1
    for p in q:
        Q *= int(p)
    for p in q[:nl]:
        Qn *= int(p)
    q_last = int(q[nl])
    worst = 0
    for k in range(0, n, 1021):
        want, _ = crt_compose([prod[l, k] for l in range(ql)], q)
        want = want - Q if want > Q // 2 else want
        got, _ = crt_compose([phase[l, k] for l in range(nl)], q[:nl])
        got = got - Qn if got > Qn // 2 else got
        worst = max(worst, abs(got - want // q_last))
    assert worst < 1 << 30, worst.bit_length()       # signal is ~2^36..2^38; fresh-noise x message terms ~2^22


def test_extreme_residues_and_empty_calls(gpu):
    """All-(q-1) and all-zero inputs through mod-up / inner product / mod-down / rescale (largest accumulators of
    the carry-free base-conversion MAC and of the 128-bit inner product), and empty ciphertexts."""
    import phantom_fhe_amd as P
    name, ql = "c4_bfv15", 30
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    top = lambda ps: np.stack([np.full(n, int(q) - 1, dtype=np.uint64) for q in ps])
    evk = np.stack([np.stack([top(primes), top(primes)]) for _ in range(size_q // size_p)])
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    for scheme in (O.CKKS, O.BFV):
        for c2 in (top(primes[:ql]), np.zeros((ql, n), dtype=np.uint64)):
            ct = np.stack([top(primes[:ql]), top(primes[:ql])])
            d_ct = P.to_device(ct, gpu)
            ctx.keyswitch_inplace(ql, d_ct, P.to_device(c2, gpu), rlk.public_keys_ptr, scheme)
            ref = tool.keyswitch_inplace(ct, c2, [evk[i] for i in range(tool.beta)], scheme)
            assert np.array_equal(P.to_host(d_ct), ref)
    ct = np.stack([top(primes[:ql]) for _ in range(2)])
    dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
    ctx.divide_and_round_q_last_ntt(ql, P.to_device(ct, gpu), 2, dst)
    assert np.array_equal(P.to_host(dst), tool.rescale_ntt(ct, 2))
    before = P.to_host(dst).copy()
    ctx.divide_and_round_q_last_ntt(ql, P.to_device(ct, gpu), 0, dst)     # empty ciphertext: nothing happens
    assert np.array_equal(P.to_host(dst), before)
    with pytest.raises(ValueError):
        ctx.divide_and_round_q_last_ntt(1, P.to_device(ct, gpu), 2, dst)  # no modulus left to drop
    with pytest.raises(ValueError):
        ctx.keyswitch_inplace(size_q + 1, P.to_device(ct, gpu), P.to_device(ct[0], gpu), rlk.public_keys_ptr, O.CKKS)


@pytest.mark.parametrize("name,scheme,ql,batch", [("hyb12_a2", O.CKKS, 6, 3), ("hyb12_a2", O.CKKS, 5, 2), ("hyb13_a3", O.CKKS, 7, 4),
                                                  ("hyb12_a2", O.BFV, 6, 2), ("hyb12_a2", O.BGV, 6, 2), ("c1_bfv4096", O.CKKS, 2, 3),
                                                  ("c4_bfv15", O.BFV, 30, 4), ("c3_ckks16", O.CKKS, 45, 2),
                                                  ("c3_ckks16", O.CKKS, 45, 5),    # 10 polynomials at N = 2^16: the r04 batched contiguous pass, mod-down epilogue
                                                  # r05: alpha = 12 at N = 2^16 -- in the experiments library (tests/test_gpu_ntt_variants.py) these take
                                                  # modup_conv_s1_kernel<., 16>: full digits, a one-limb last digit, the BGV branch
                                                  ("hyb16_a12", O.CKKS, 24, 2), ("hyb16_a12", O.CKKS, 13, 2), ("hyb16_a12", O.BGV, 24, 2)])
def test_batched_keyswitch_and_tensor(name, scheme, ql, batch, gpu):
    """pha_keyswitch_inplace_batched / pha_tensor_prod_2x2_batched: every ciphertext of the batch must equal the
    single-ciphertext result (oracle), including short last digits and the alpha = 1 path."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        ctx.set_plain_modulus(BGV_T)
        tool.set_plain_modulus(BGV_T)
    r = rng_for(110)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    ct1 = np.stack([np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)]) for _ in range(batch)])
    ct2 = np.stack([np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)]) for _ in range(batch)])
    d01 = P.to_device(ct1, gpu)
    d2 = P.to_device(np.zeros((batch, ql, n), dtype=np.uint64), gpu)
    ctx.tensor_prod_2x2_batched(d01, P.to_device(ct2, gpu), d01, d2, ql, batch)     # in place on operand 1
    ref3 = [oc.tensor_prod_2x2(ct1[b], ct2[b], ql) for b in range(batch)]
    assert np.array_equal(P.to_host(d01), np.stack([t[:2] for t in ref3]))
    assert np.array_equal(P.to_host(d2), np.stack([t[2] for t in ref3]))
    ctx.keyswitch_inplace_batched(ql, d01, d2, batch, rlk.public_keys_ptr, scheme)
    got = P.to_host(d01)
    for b in range(batch):
        ref = tool.keyswitch_inplace(ref3[b][:2], ref3[b][2], [evk[i] for i in range(tool.beta)], scheme)
        assert np.array_equal(got[b], ref), f"ciphertext {b}"
    if scheme == O.CKKS and ql > 1:
        out = P.to_device(np.zeros((batch, 2, ql - 1, n), dtype=np.uint64), gpu)
        ctx.divide_and_round_q_last_ntt(ql, d01, 2 * batch, out)                     # rescale of the whole batch
        assert np.array_equal(P.to_host(out).reshape(2 * batch, ql - 1, n), tool.rescale_ntt(got.reshape(2 * batch, ql, n), 2 * batch))
    ctx.keyswitch_inplace_batched(ql, d01, d2, 0, rlk.public_keys_ptr, scheme)       # empty batch: no-op


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("hyb12_a2", 65537), ("hyb12_a2", 1032193),
                                          ("hyb13_a3", 786433), ("c4_bfv15", 1032193)])
def test_bfv_multiply_behz(name, plain_t, gpu):
    """bfv_multiply_behz (src/evaluate.cu:447-548) incl. the squaring path, vs the oracle; at C4 the conversions
    run the wide (31-input) form."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(120)
    ct1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    dst = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    with pytest.raises(ValueError):
        ctx.bfv_multiply_behz(P.to_device(ct1, gpu), P.to_device(ct2, gpu), dst)    # no plain modulus yet
    ctx.set_plain_modulus(plain_t)
    behz = O.Behz(oc, plain_t)
    d1, d2 = P.to_device(ct1, gpu), P.to_device(ct2, gpu)
    ctx.bfv_multiply_behz(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), behz.multiply(ct1, ct2))
    ctx.bfv_multiply_behz(d1, d1, dst)                                               # square
    assert np.array_equal(P.to_host(dst), behz.multiply(ct1, ct1))
    # the key-switch path still works after the table arrays grew (relinearize the product)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    tool = O.Tool(oc, size_q)
    prod = behz.multiply(ct1, ct1)
    d_ct = P.to_device(prod[:2], gpu)
    ctx.keyswitch_inplace(size_q, d_ct, P.to_device(prod[2], gpu), rlk.public_keys_ptr, O.BFV)
    assert np.array_equal(P.to_host(d_ct), tool.keyswitch_inplace(prod[:2], prod[2], [evk[i] for i in range(tool.beta)], O.BFV))


def test_bfv_hommul_relinearize_decrypts_on_gpu(gpu):
    """Config 0's operation (BFV HomAdd + HomMul at N = 4096, BFVDefault primes) on the GPU path: encryptions of m1, m2
    -> add, BEHZ multiply, relinearize with a key generated by this build -> decrypts to m1 + m2 and m1 * m2 mod t."""
    import phantom_fhe_amd as P
    from util import crt_compose
    name, plain_t = "c1_bfv4096", 65537
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    q = [int(p) for p in primes[:size_q]]
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    r = rng_for(130)
    Q = 1
    for p in q:
        Q *= p
    delta = Q // plain_t
    s_small = r.integers(-1, 2, n)
    sk_ntt_full = _ternary_sk_from(oc, s_small, primes)
    sk_ntt = sk_ntt_full[:size_q]
    d_sk = P.to_device(sk_ntt_full, gpu)
    d_s2 = P.to_device(np.zeros((size_q, n), dtype=np.uint64), gpu)
    ctx.multiply_rns_poly(d_sk, d_sk, d_s2, size_q)
    dnum = size_q // size_p
    rlk = ctx.generate_one_kswitch_key(d_sk, d_s2, P.to_device(np.stack([uniform_poly(r, primes, n) for _ in range(dnum)]), gpu),
                                       P.to_device(_noise(r, primes, n, dnum), gpu), O.BFV)

    def encrypt(m):
        a = uniform_poly(r, q, n)
        e = r.integers(-3, 4, n)
        dm = np.stack([np.array([(delta * int(v) + int(ev)) % p for v, ev in zip(m, e)], dtype=np.uint64) for p in q])
        a_s = oc.nwt_backward(oc.multiply(oc.nwt_forward(a, size_q, 0), sk_ntt, size_q), size_q)
        return np.stack([oc.sub(dm, a_s, size_q), a])

    def decrypt(ct):      # ct [2][Q][N] coefficient form -> plaintext coefficients mod t
        c1s = oc.nwt_backward(oc.multiply(oc.nwt_forward(ct[1], size_q, 0), sk_ntt, size_q), size_q)
        phase = oc.add(ct[0], c1s, size_q)
        out = []
        for k in range(0, n, 53):
            v, _ = crt_compose([phase[l, k] for l in range(size_q)], q)
            out.append(((v * plain_t + Q // 2) // Q) % plain_t)
        return out

    m1 = r.integers(0, plain_t, n)
    m2 = r.integers(0, plain_t, n)
    c1, c2 = encrypt(m1), encrypt(m2)
    d1, d2 = P.to_device(c1, gpu), P.to_device(c2, gpu)
    # HomAdd
    dsum = P.to_device(np.zeros_like(c1), gpu)
    for i in range(2):
        ctx.add_rns_poly(d1[i], d2[i], dsum[i], size_q)
    assert decrypt(P.to_host(dsum)) == [int((int(m1[k]) + int(m2[k])) % plain_t) for k in range(0, n, 53)]
    # HomMul + relinearize
    d3 = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    ctx.bfv_multiply_behz(d1, d2, d3)
    ctx.keyswitch_inplace(size_q, d3[:2], d3[2], rlk.public_keys_ptr, O.BFV)
    big = int(O.get_primes(n, 60, 1)[0])
    bc = O.Ctx(log_n, [big], 0)
    pm = bc.nwt_backward(bc.multiply(bc.nwt_forward(m1.astype(np.uint64).reshape(1, n), 1, 0),
                                     bc.nwt_forward(m2.astype(np.uint64).reshape(1, n), 1, 0), 1), 1)[0]
    want = []
    for k in range(0, n, 53):
        w = int(pm[k])
        w = w - big if w > big // 2 else w
        want.append(w % plain_t)
    assert decrypt(P.to_host(d3[:2].contiguous())) == want


def _ternary_sk_from(oc, s_small, primes):
    n = len(s_small)
    sk = np.stack([(s_small % int(p)).astype(np.uint64) for p in primes])
    return oc.nwt_forward(sk, len(primes), 0)


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("bfv13_50", 65537), ("bfv13_50", 1032193), ("c4_bfv15", 1032193)])
def test_bfv_multiply_hps(name, plain_t, gpu):
    """bfv_multiply_hps, mul_tech hps (src/evaluate.cu:674-818) incl. the squaring path, vs the oracle (same fma chains,
    same in-place carry reduction)."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    hps = O.Hps(oc, plain_t)
    r = rng_for(140)
    ct1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2[:, :, :32] = np.array(primes[:size_q], dtype=np.uint64)[None, :, None] - 1       # extreme residues
    dst = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    d1, d2 = P.to_device(ct1, gpu), P.to_device(ct2, gpu)
    ctx.bfv_multiply_hps(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), hps.multiply(ct1, ct2))
    ctx.bfv_multiply_hps(d2, d2, dst)
    assert np.array_equal(P.to_host(dst), hps.multiply(ct2, ct2))
    # both variants on one context (two sets of auxiliary table rows)
    ctx.bfv_multiply_behz(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.Behz(oc, plain_t).multiply(ct1, ct2))
    ctx.bfv_multiply_hps(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), hps.multiply(ct1, ct2))


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("bfv13_50", 65537), ("hyb12_a2", 1032193), ("c4_bfv15", 1032193)])
def test_bfv_multiply_hps_overq(name, plain_t, gpu):
    """bfv_multiply_hps with mul_tech hps_overq (src/evaluate.cu:674-818, overq branches), incl. the reference's
    squaring shortcut (same buffer on both sides), vs the oracle; then hps on the same context (shared base R rows)."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    hq = O.HpsOverQ(oc, plain_t)
    r = rng_for(141)
    ct1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2[:, :, :32] = np.array(primes[:size_q], dtype=np.uint64)[None, :, None] - 1       # extreme residues
    ct2[:, :, 32:64] = 0
    dst = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    d1, d2 = P.to_device(ct1, gpu), P.to_device(ct2, gpu)
    ctx.bfv_multiply_hps_overq(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), hq.multiply(ct1, ct2))
    ctx.bfv_multiply_hps_overq(d2, d1, dst)
    assert np.array_equal(P.to_host(dst), hq.multiply(ct2, ct1))          # not symmetric: only the second operand is switched
    ctx.bfv_multiply_hps_overq(d2, d2, dst)
    assert np.array_equal(P.to_host(dst), hq.multiply(ct2, ct2))          # squaring shortcut
    ctx.bfv_multiply_hps(d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.Hps(oc, plain_t).multiply(ct1, ct2))
    assert np.array_equal(P.to_host(d1), ct1) and np.array_equal(P.to_host(d2), ct2)


@pytest.mark.parametrize("name,plain_t,ql", [("bfv13_50", 65537, 3), ("bfv13_50", 65537, 2), ("hyb12_a2", 1032193, 4),
                                             ("c4_bfv15", 1032193, 29), ("c4_bfv15", 1032193, 15)])
def test_bfv_hps_overq_leveled(name, plain_t, ql, gpu):
    """hps_overq_leveled with levels dropped: scaleAndRound_HPS_Q_Ql, ExpandCRTBasis_Ql_Q (src/rns.cu:1798-1836), the
    leveled multiply (src/evaluate.cu:709-711,747-748,794-795) and the leveled BFV key switch
    (src/eval_key_switch.cu:142-147,170-175) vs the oracle."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    hq = O.HpsOverQ(oc, plain_t, ql)
    r = rng_for(150 + ql)
    ct1 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    ct2[:, :, :32] = np.array(primes[:size_q], dtype=np.uint64)[None, :, None] - 1
    d1, d2 = P.to_device(ct1, gpu), P.to_device(ct2, gpu)
    low = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.scaleAndRound_HPS_Q_Ql(ql, low, d2[0])
    y = hq.scale_q_ql(ct2[0])
    assert np.array_equal(P.to_host(low), y)
    full = P.to_device(np.full((size_q, n), 5, dtype=np.uint64), gpu)
    ctx.ExpandCRTBasis_Ql_Q(ql, full, low)
    assert np.array_equal(P.to_host(full), hq.expand_ql_q(y))
    dst = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    ctx.bfv_multiply_hps_overq_leveled(ql, d1, d2, dst)
    ref = hq.multiply(ct1, ct2)
    assert np.array_equal(P.to_host(dst), ref)
    ctx.bfv_multiply_hps_overq_leveled(ql, d2, d2, dst)
    assert np.array_equal(P.to_host(dst), hq.multiply(ct2, ct2))
    # leveled relinearisation of the product
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    tool = O.Tool(oc, ql)
    d_ct = P.to_device(ref[:2], gpu)
    ctx.keyswitch_inplace_bfv_leveled(ql, d_ct, P.to_device(ref[2], gpu), rlk.public_keys_ptr)
    assert np.array_equal(P.to_host(d_ct), hq.keyswitch_leveled(tool, ref[:2], ref[2], [evk[i] for i in range(tool.beta)]))
    fused = P.to_device(np.zeros((2, size_q, n), dtype=np.uint64), gpu)
    ctx.bfv_mul_relin_hps_overq_leveled(ql, d1, d2, rlk.public_keys_ptr, fused)
    assert np.array_equal(P.to_host(fused), hq.mul_relin_leveled(tool, ct1, ct2, [evk[i] for i in range(tool.beta)]))
    with pytest.raises(ValueError):
        ctx.scaleAndRound_HPS_Q_Ql(size_q, low, d2[0])          # nothing dropped
    # the top level through the leveled entry is plain hps_overq
    ctx.bfv_multiply_hps_overq_leveled(size_q, d1, d2, dst)
    assert np.array_equal(P.to_host(dst), O.HpsOverQ(oc, plain_t).multiply(ct1, ct2))


@pytest.mark.parametrize("bits,m,n,k,batch", [(50, 256, 256, 256, 3), (50, 96, 40, 72, 2), (60, 128, 64, 100, 2), (36, 64, 32, 16, 1),
                                                 (60, 64, 32, 1200, 2), (50, 128, 64, 64, 9), (50, 64, 192, 96, 2), (60, 64, 64, 256, 3),
                                                 (50, 1, 1, 1, 1), (50, 65, 33, 33, 2), (49, 64, 64, 4096, 1)])
def test_batched_modular_gemm(bits, m, n, k, batch, gpu):
    """matmul_bench shape (256^3 per 50-bit modulus) and ragged shapes, wide (60-bit) and narrow paths, vs the oracle;
    the benchmark's all-ones input gives k everywhere."""
    import phantom_fhe_amd as P
    primes = [int(p) for p in O.get_primes(4096, bits, batch)]
    ctx = P.PhantomContext(12, primes, 0, device=gpu)
    r = rng_for(150)
    A = np.stack([r.integers(0, q, (m, k), dtype=np.uint64) for q in primes])
    B = np.stack([r.integers(0, q, (k, n), dtype=np.uint64) for q in primes])
    A[:, 0, :] = np.array(primes, dtype=np.uint64)[:, None] - 1       # a row and a column of q - 1
    B[:, :, 0] = np.array(primes, dtype=np.uint64)[:, None] - 1
    if k > 256:   # 60-bit residues just below q everywhere: k q^2 exceeds 2^128 unless the totals are reduced on the way
        A = (np.array(primes, dtype=np.uint64)[:, None, None] - 1 - r.integers(0, 4, A.shape, dtype=np.uint64))
        B = (np.array(primes, dtype=np.uint64)[:, None, None] - 1 - r.integers(0, 4, B.shape, dtype=np.uint64))
    dC = P.to_device(np.zeros((batch, m, n), dtype=np.uint64), gpu)
    ctx.batched_modular_gemm(dC, P.to_device(A, gpu), P.to_device(B, gpu), m, n, k, batch)
    got = P.to_host(dC)
    for z, q in enumerate(primes):
        assert np.array_equal(got[z], O.gemm_mod(q, A[z], B[z])), f"modulus {z}"
    ones_a, ones_b = np.ones_like(A), np.ones_like(B)
    ctx.batched_modular_gemm(dC, P.to_device(ones_a, gpu), P.to_device(ones_b, gpu), m, n, k, batch)
    assert (P.to_host(dC) == k).all()


@pytest.mark.parametrize("bits,x", [(50, 0x808080808080), (50, 0x7f7f7f7f7f7f), (60, 0x0080808080808080), (60, 0x007f7f7f7f7f7f7f)])
def test_batched_modular_gemm_largest_digits_longest_k(bits, x, gpu):
    """Headroom of the int32 digit-plane accumulators: every operand has the extreme signed bytes (-128 / +127 in all but the top
    digit) and k is the largest the entry accepts (16384), so each accumulator holds 7 (8) x 16384 products of magnitude 2^14; the
    exact answer is k x^2 mod q everywhere."""
    import phantom_fhe_amd as P
    primes = [int(p) for p in O.get_primes(4096, bits, 2)]
    assert all(x < q for q in primes)
    ctx = P.PhantomContext(12, primes, 0, device=gpu)
    m, n, k = 64, 96, 16384
    A = P.to_device(np.full((2, m, k), x, dtype=np.uint64), gpu)
    B = P.to_device(np.full((2, k, n), x, dtype=np.uint64), gpu)
    dC = P.to_device(np.zeros((2, m, n), dtype=np.uint64), gpu)
    ctx.batched_modular_gemm(dC, A, B, m, n, k, 2)
    got = P.to_host(dC)
    for z, q in enumerate(primes):
        assert (got[z] == np.uint64(k * x * x % q)).all(), f"modulus {z}"


@pytest.mark.parametrize("name,ibase,obase", [
    ("hyb12_a2", [6, 7], [0, 1, 2, 3, 4, 5]),            # P -> Q (2 inputs)
    ("hyb12_a2", [0, 1, 2, 3, 4, 5], [6, 7]),            # Q -> P
    ("hyb12_a2", [1], [0, 2, 7]),                        # a single input prime
    ("hyb13_a3", [3, 4, 5], [0, 1, 2, 6, 7, 8, 9, 10, 11]),
    ("c4_bfv15", list(range(30)), list(range(30, 45))),  # 30 inputs (register-resident path)
    ("c3_ckks16", list(range(45)), list(range(45, 60))), # 45 inputs (wide path)
    ("c3_ckks16", list(range(45, 60)), list(range(45))), # the mod-down conversion at C3
])
def test_base_converter_object(name, ibase, obase, gpu):
    """DBaseConverter (include/rns_bconv.cuh:13-87) for arbitrary bases: bConv_BEHZ (rns_bconv.cu:212-229) and
    bConv_HPS (:248-372) vs the oracle."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    ctx = _ctx(name, gpu)
    ip, op = [primes[i] for i in ibase], [primes[j] for j in obase]
    r = rng_for(1200 + len(ibase))
    src = uniform_poly(r, ip, n)
    src[:, :8] = 0
    src[:, 8:16] = np.array(ip, dtype=np.uint64)[:, None] - 1
    conv = P.DBaseConverter(ctx, ibase, obase)
    dst = P.to_device(np.zeros((len(obase), n), dtype=np.uint64), gpu)
    conv.bConv_BEHZ(dst, P.to_device(src, gpu))
    assert np.array_equal(P.to_host(dst), O.bconv(ip, op, src, n))
    conv.bConv_HPS(dst, P.to_device(src, gpu))
    hps = O.bconv_hps(ip, op, src, n)
    assert np.array_equal(P.to_host(dst), hps)
    if len(ibase) <= 3:      # the HPS conversion is exact: x reduced mod p_j, x the centred CRT lift of the residues
        Q = 1
        for q in ip:
            Q *= int(q)
        for k in range(0, n, 331):
            x, _ = crt_compose([src[i, k] for i in range(len(ip))], ip)
            x = x - Q if x > Q // 2 else x
            assert all(int(hps[j, k]) == x % int(p) for j, p in enumerate(op))


def test_base_converter_rejects_bad_bases(gpu):
    import phantom_fhe_amd as P
    ctx = _ctx("hyb12_a2", gpu)
    with pytest.raises(ValueError):
        P.DBaseConverter(ctx, [0, 0], [1])
    with pytest.raises(ValueError):
        P.DBaseConverter(ctx, [0], [99])
    with pytest.raises(ValueError):
        P.DBaseConverter(ctx, [], [1])


def test_concurrent_host_threads_on_their_own_streams(gpu):
    """The context is immutable after construction and its scratch arenas are per stream, so several host threads may
    evaluate on their own streams at once (the reference is built --default-stream per-thread for the same use)."""
    import threading
    import torch
    import phantom_fhe_amd as P
    name = "hyb13_a3"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(160)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    jobs = []
    for i, ql in enumerate((9, 7, 4, 9)):       # different levels: the per-level tools are built concurrently too
        ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
        c2 = uniform_poly(r, primes[:ql], n)
        tool = O.Tool(oc, ql)
        want = tool.keyswitch_inplace(ct, c2, [evk[k] for k in range(tool.beta)], O.CKKS)
        want = tool.rescale_ntt(want, 2) if ql > 1 else want
        jobs.append((ql, ct, c2, want))
    results, errors = [None] * len(jobs), []

    def work(i):
        try:
            ql, ct, c2, _ = jobs[i]
            stream = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(stream):
                for _ in range(5):                                   # repeat: reuse of the stream's arena
                    d_ct, d_c2 = P.to_device(ct, gpu), P.to_device(c2, gpu)
                    ctx.keyswitch_inplace(ql, d_ct, d_c2, rlk.public_keys_ptr, O.CKKS)
                    dst = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
                    ctx.divide_and_round_q_last_ntt(ql, d_ct, 2, dst)
                stream.synchronize()
                results[i] = P.to_host(dst)
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (_, _, _, want) in enumerate(jobs):
        assert np.array_equal(results[i], want)


def test_short_lived_threads_on_the_default_stream_give_their_arenas_back(gpu):
    """The null / per-thread stream handles name a different stream in every host thread, so each such thread gets its own
    scratch arena -- and returns it when it exits (a thread pool must not grow one arena per thread for the life of the context)."""
    import threading
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(161)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    tool = O.Tool(oc, size_q)
    ct = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    c2 = uniform_poly(r, primes[:size_q], n)
    want = tool.keyswitch_inplace(ct, c2, [evk[k] for k in range(tool.beta)], O.CKKS)
    errors, counts = [], []

    def work():
        try:
            d_ct, d_c2 = P.to_device(ct, gpu), P.to_device(c2, gpu)        # torch's default stream = the null stream
            ctx.keyswitch_inplace(size_q, d_ct, d_c2, rlk.public_keys_ptr, O.CKKS)
            assert np.array_equal(P.to_host(d_ct), want)
            counts.append(ctx.arena_count())
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)

    for _ in range(6):             # one after the other: each thread is gone before the next starts
        t = threading.Thread(target=work)
        t.start()
        t.join()
    assert not errors, errors
    assert max(counts) <= 2 and ctx.arena_count() <= 1, (counts, ctx.arena_count())


def test_key_switch_and_rescale_replay_from_a_hip_graph(gpu):
    """The entry points only enqueue work on the caller's stream once the stream's scratch arena has its size, so a
    HomMul -> relinearize -> rescale sequence can be captured into a hipGraph and replayed (C3: 501 us per replay vs
    496 us eager: the sequence is GPU-bound, the graph only removes host work)."""
    import torch
    import phantom_fhe_amd as P
    name, ql = "hyb13_a3", 9
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(170)
    evk = _keys(oc, r, primes, n, size_q, size_p)
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    a = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    b = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_a, d_b = P.to_device(a, gpu), P.to_device(b, gpu)
    buf = P.to_device(np.zeros((3, ql, n), dtype=np.uint64), gpu)
    out = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)

    def hommul():
        buf[:2].copy_(d_a)
        ctx.tensor_prod_2x2_rns_poly(buf, d_b, buf, ql)
        ctx.keyswitch_inplace(ql, buf, buf[2], rlk.public_keys_ptr, O.CKKS)
        ctx.divide_and_round_q_last_ntt(ql, buf, 2, out)

    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):
        hommul()                              # warm-up on the capture stream: its arena is allocated here
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        hommul()
    t3 = oc.tensor_prod_2x2(a, b, ql)
    want = tool.rescale_ntt(tool.keyswitch_inplace(t3[:2], t3[2], [evk[i] for i in range(tool.beta)], O.CKKS), 2)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(P.to_host(out), want)


@pytest.mark.parametrize("ntt_form", [False, True])
def test_apply_galois_batched(ntt_form, gpu):
    """One launch for the permutation of several polynomials == the per-polynomial calls (src/galois.cu:11-39)."""
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    ctx = _ctx(name, gpu)
    ql, polys, elt = 5, 6, 2 * n - 1
    x = np.stack([uniform_poly(rng_for(180 + p), primes[:ql], n) for p in range(polys)])
    d_x = P.to_device(x, gpu)
    ref = P.to_device(np.zeros_like(x), gpu)
    for p in range(polys):
        if ntt_form:
            ctx.apply_galois_ntt(d_x[p], ref[p], elt, ql)
        else:
            ctx.apply_galois(d_x[p], ref[p], elt, ql)
    out = P.to_device(np.zeros_like(x), gpu)
    ctx.apply_galois_batched(d_x, out, elt, ql, polys, ntt_form)
    assert np.array_equal(P.to_host(out), P.to_host(ref))


def test_empty_batches_and_zero_sizes_are_no_ops(gpu):
    """Zero-sized work (no ciphertexts in a batch, no polynomials, no limbs) returns without touching the buffers."""
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    r = rng_for(190)
    rlk = P.PhantomRelinKey.from_numpy(_keys(oc, r, primes, n, size_q, size_p), gpu)
    x = np.stack([uniform_poly(r, primes[:size_q], n) for _ in range(2)])
    d = P.to_device(x, gpu)
    ctx.keyswitch_inplace_batched(size_q, d, d[0], 0, rlk.public_keys_ptr, O.CKKS)
    ctx.tensor_prod_2x2_batched(d, d, d, d[0], size_q, 0)
    ctx.apply_galois_batched(d, d[1:], 3, size_q, 0, True)
    ctx.divide_and_round_q_last_ntt(size_q, d, 0, d)
    ctx.add_rns_poly(d, d, d, 0)
    ctx.nwt_2d_radix8_forward_inplace_batched(d, 0, 0, 2, size_q * n)
    assert np.array_equal(P.to_host(d), x)
    with pytest.raises(ValueError):
        ctx.keyswitch_inplace_batched(size_q, d, d[0], 2000, rlk.public_keys_ptr, O.CKKS)     # beyond the batch limit
