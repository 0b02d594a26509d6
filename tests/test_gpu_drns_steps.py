"""The launchers SURVEY 8(b) / row f3 cite that r05 had only inside whole operations (VERDICT r05 "What's missing" 1-2), each
through the C ABI against the oracle's restatement of the same reference lines:

  DRNSTool::moddown                     include/rns.cuh:159-160, src/rns_bconv.cu:712-761
  DBaseConverter::bConv_BEHZ_var1       include/rns_bconv.cuh:64, src/rns_bconv.cu:231-246
  DBaseConverter::exact_convert_array   include/rns_bconv.cuh:68, src/rns_bconv.cu:374-431
  DRNSTool::fastbconv_m_tilde / sm_mrq / fast_floor / fastbconv_sk                           src/rns.cu:1249-1510
  DRNSTool::scaleAndRound_HPS_QR_R / scaleAndRound_HPS_QlRl_Ql / ExpandCRTBasis_Ql_Q_add_to_ct   src/rns.cu:1700-1858

and a caller written like the reference's own (bfv_multiply_behz src/evaluate.cu:404-548, the HPS tail :790-808) that uses the
step entries only and must land on the whole multiply's words."""
import numpy as np
import pytest

from oracle import oracle as O
from util import crt_compose, oracle_ctx, primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _ctx(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    return P.PhantomContext(log_n, list(primes), size_p, device=gpu)


def _extremes(x, primes):
    """The first coefficients of every limb at 0 and at q - 1."""
    x[:, :8] = 0
    x[:, 8:16] = np.array(primes, dtype=np.uint64)[:, None] - 1
    return x


MODDOWN_CASES = [
    ("hyb12_a2", O.BFV, 6, None), ("hyb12_a2", O.CKKS, 6, None), ("hyb12_a2", O.CKKS, 3, None), ("hyb12_a2", O.BGV, 6, 65537),
    ("hyb13_a3", O.BFV, 7, None),                      # short last digit at that level (the tool's own constants)
    ("c1_bfv4096", O.BFV, 2, None), ("c1_bfv4096", O.CKKS, 2, None), ("c1_bfv4096", O.BGV, 2, 65537),   # alpha = 1: no special kernel in moddown (:733, :746)
    ("c4_bfv15", O.BFV, 30, None), ("c4_bfv15", O.BGV, 30, 786433),
    ("c3_ckks16", O.CKKS, 45, None), ("c3_ckks16", O.CKKS, 31, None),
    ("wide_p33", O.BFV, 6, None),                      # alpha > 32: conversion into delta + the element-wise kernel
    ("p61_a2", O.BFV, 6, None),
]


@pytest.mark.parametrize("name,scheme,ql,plain_t", MODDOWN_CASES)
def test_moddown(name, scheme, ql, plain_t, gpu):
    """DRNSTool::moddown (src/rns_bconv.cu:712-761) for the three schemes: BFV input in COEFFICIENT form (no inverse transform),
    CKKS / BGV in NTT form.  Also: moddown(bfv, x) == moddown_from_NTT(bfv, NTT(x)), the relation between the two launchers."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if plain_t:
        ctx.set_plain_modulus(plain_t)
        tool.set_plain_modulus(plain_t)
    qlp = list(primes[:ql]) + list(primes[size_q:])
    cx = _extremes(uniform_poly(rng_for(2000 + ql), qlp, n), qlp)
    want = tool.moddown(cx, scheme)
    d_cx = P.to_device(cx, gpu)
    d_ct = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.moddown(ql, d_ct, d_cx, scheme)
    assert np.array_equal(P.to_host(d_ct), want)
    if scheme == O.BFV:
        # the other launcher on the transformed input gives the same words; on the SAME input it does not
        ntt = oc.nwt_forward_map(cx, list(range(ql)) + list(range(size_q, size_q + size_p)))
        d_cx = P.to_device(ntt, gpu)
        ctx.moddown_from_NTT(ql, d_ct, d_cx, scheme)
        assert np.array_equal(P.to_host(d_ct), want)
        assert not np.array_equal(tool.moddown_from_ntt(cx, scheme), want)
    else:
        # CKKS / BGV: same input convention, same stored words as moddown_from_NTT (fused epilogue == separate kernels)
        assert np.array_equal(tool.moddown_from_ntt(cx, scheme), want)


@pytest.mark.parametrize("name,ibase,obase", [
    ("hyb12_a2", [0, 1, 2, 3, 4, 5], [6, 7]),
    ("hyb12_a2", [6, 7], [0, 1, 2, 3, 4, 5]),
    ("hyb12_a2", [1], [0, 2, 7]),
    ("hyb13_a3", [3, 4, 5], [0, 1, 2, 6, 7, 8, 9, 10, 11]),
    ("c4_bfv15", list(range(30)), list(range(30, 45))),      # the shape of base_Q_to_Rl_conv at config 4 (30 inputs)
    ("c3_ckks16", list(range(45)), list(range(45, 60))),     # 45 inputs: the wide path
])
def test_bconv_behz_var1(name, ibase, obase, gpu):
    """bConv_BEHZ_var1 (src/rns_bconv.cu:231-246, constants src/host/rns.cu:469-496) vs the oracle, and vs its definition in Python
    integers on sampled coefficients: out_j = sum_i [x_i * (-P qhat_i^-1) mod q_i] * (q_i^-1 mod p_j) mod p_j, P = prod(obase)."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    ctx = _ctx(name, gpu)
    ip, op = [int(primes[i]) for i in ibase], [int(primes[j]) for j in obase]
    src = _extremes(uniform_poly(rng_for(2100 + len(ibase)), ip, n), ip)
    conv = P.DBaseConverter(ctx, ibase, obase)
    dst = P.to_device(np.zeros((len(obase), n), dtype=np.uint64), gpu)
    conv.bConv_BEHZ_var1(dst, P.to_device(src, gpu))
    got = P.to_host(dst)
    assert np.array_equal(got, O.bconv_behz_var1(ip, op, src, n))
    big_p = 1
    for p in op:
        big_p *= p
    for k in list(range(0, 16)) + list(range(16, n, 509)):
        y = []
        for i, qi in enumerate(ip):
            hat = 1
            for m, qm in enumerate(ip):
                if m != i:
                    hat = hat * qm % qi
            y.append(int(src[i, k]) * ((-big_p * pow(hat, -1, qi)) % qi) % qi)
        for j, pj in enumerate(op):
            assert int(got[j, k]) == sum(yi * pow(qi, -1, pj) for yi, qi in zip(y, ip)) % pj
    # the plain conversion of the same object still gives bConv_BEHZ (the var1 constants live beside it)
    conv.bConv_BEHZ(dst, P.to_device(src, gpu))
    assert np.array_equal(P.to_host(dst), O.bconv(ip, op, src, n))


@pytest.mark.parametrize("name,ibase,t", [
    ("hyb12_a2", [0, 1, 2, 3, 4, 5], 65537), ("hyb12_a2", [0, 1], 1 << 20), ("hyb12_a2", [3], 1032193),
    ("c1_bfv4096", [0, 1], 1032193),                 # base_q_to_t_conv_ of config 1 (hps_decrypt / decrypt path's shape)
    ("c4_bfv15", list(range(30)), 786433),
    ("c3_ckks16", list(range(45)), (1 << 60) + 33),  # a wide, composite modulus
])
def test_exact_convert_array(name, ibase, t, gpu):
    """exact_convert_array (src/rns_bconv.cu:374-431) to ONE raw modulus vs the oracle; for short bases also vs the definition:
    the centred CRT lift of the residues, reduced modulo t."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    ctx = _ctx(name, gpu)
    ip = [int(primes[i]) for i in ibase]
    src = _extremes(uniform_poly(rng_for(2200 + len(ibase)), ip, n), ip)
    conv = P.DBaseConverter(ctx, ibase, out_modulus=t)
    dst = P.to_device(np.zeros(n, dtype=np.uint64), gpu)
    conv.exact_convert_array(dst, P.to_device(src, gpu))
    got = P.to_host(dst)
    assert np.array_equal(got, O.exact_convert_array(ip, t, src, n))
    if len(ip) <= 2:
        Q = 1
        for q in ip:
            Q *= q
        for k in range(0, n, 97):
            x, _ = crt_compose([src[i, k] for i in range(len(ip))], ip)
            x = x - Q if 2 * x > Q else x
            assert int(got[k]) == x % t
    with pytest.raises(ValueError):
        conv.bConv_BEHZ(dst, P.to_device(src, gpu))              # a raw-modulus converter converts exactly, nothing else
    with pytest.raises(ValueError):
        P.DBaseConverter(ctx, ibase, out_modulus=1)
    # a table-row converter with ONE output row serves it too; more rows are refused as the reference does (:423-425)
    if name == "hyb12_a2" and len(ibase) == 2:
        row = P.DBaseConverter(ctx, ibase, [7])
        row.exact_convert_array(dst, P.to_device(src, gpu))
        assert np.array_equal(P.to_host(dst), O.exact_convert_array(ip, int(primes[7]), src, n))
        with pytest.raises(ValueError, match="must be one"):
            P.DBaseConverter(ctx, ibase, [6, 7]).exact_convert_array(dst, P.to_device(src, gpu))


def _aux_row0(ctx, first_prime, start):
    """Row of the context's prime table that holds `first_prime` (the auxiliary bases follow the QP primes)."""
    row = start
    while True:     # prime_info raises ValueError past the last row
        if ctx.prime_info(row)["value"] == int(first_prime):
            return row
        row += 1


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("hyb12_a2", 1032193), ("hyb13_a3", 786433), ("c4_bfv15", 1032193)])
def test_behz_steps(name, plain_t, gpu):
    """The four BEHZ steps one by one vs the oracle, then bfv_multiply_behz written as the reference's caller writes it
    (src/evaluate.cu:404-548) on the step entries + the NTT / dyadic entries: the whole multiply's words, bit for bit."""
    import torch
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    behz = O.Behz(oc, plain_t)
    sk = behz.size_bsk
    assert ctx.tool_aux_sizes(size_q)[0] == sk
    r = rng_for(2300)
    qp = list(primes[:size_q])
    zeros = lambda limbs: P.to_device(np.zeros((limbs, n), dtype=np.uint64), gpu)   # noqa: E731

    # ---- each step against the oracle on its own seeded input ----
    x_q = _extremes(uniform_poly(r, qp, n), qp)
    lift = zeros(sk + 1)
    ctx.fastbconv_m_tilde(size_q, lift, P.to_device(x_q, gpu))
    want_lift = behz.fastbconv_m_tilde(x_q)
    assert np.array_equal(P.to_host(lift), want_lift) and int(want_lift[sk].max()) < (1 << 32)
    bskmt = list(behz.bsk) + [1 << 32]
    x_lift = _extremes(uniform_poly(r, bskmt, n), bskmt)
    out_bsk = zeros(sk)
    ctx.sm_mrq(size_q, out_bsk, P.to_device(x_lift, gpu))
    assert np.array_equal(P.to_host(out_bsk), behz.sm_mrq(x_lift))
    x_bsk = _extremes(uniform_poly(r, behz.bsk, n), behz.bsk)
    ctx.fast_floor(size_q, P.to_device(x_q, gpu), P.to_device(x_bsk, gpu), out_bsk)
    assert np.array_equal(P.to_host(out_bsk), behz.fast_floor(x_q, x_bsk))
    out_q = zeros(size_q)
    ctx.fastbconv_sk(size_q, P.to_device(x_bsk, gpu), out_q)
    assert np.array_equal(P.to_host(out_q), behz.fastbconv_sk(x_bsk))
    if size_q > 1:
        with pytest.raises(ValueError, match="top data level"):
            ctx.sm_mrq(size_q - 1, out_bsk, P.to_device(x_lift, gpu))

    # ---- the caller of src/evaluate.cu:404-548 on the step entries ----
    ct1 = np.stack([uniform_poly(r, qp, n) for _ in range(2)])
    ct2 = np.stack([uniform_poly(r, qp, n) for _ in range(2)])
    aux0 = _aux_row0(ctx, behz.bsk[0], len(primes))

    def behz_mul_1(ct):   # BEHZ_mul_1 :404-441
        enc_q = P.to_device(ct, gpu)
        enc_bsk = P.to_device(np.zeros((2, sk, n), dtype=np.uint64), gpu)
        tmp = zeros(sk + 1)
        for i in range(2):
            src = P.to_device(ct[i], gpu)
            ctx.nwt_2d_radix8_forward_inplace(enc_q[i], size_q, 0)
            ctx.fastbconv_m_tilde(size_q, tmp, src)
            ctx.sm_mrq(size_q, enc_bsk[i], tmp)
            ctx.nwt_2d_radix8_forward_inplace_include_temp_mod(enc_bsk[i], sk, 0, sk + 1)
        return enc_q, enc_bsk

    def tensor(a, b, limbs, start):   # tensor_prod_2x2_rns_poly<<<>>>(.., base_rns | base_Bsk, ..) :479-498
        res = P.to_device(np.zeros((3, limbs, n), dtype=np.uint64), gpu)
        ctx.tensor_prod_2x2_rns_poly_at(a, b, res, limbs, start)
        # the dyadic entries over the same rows store the same canonical words
        alt = P.to_device(np.zeros((3, limbs, n), dtype=np.uint64), gpu)
        ctx.multiply_rns_poly(a[0], b[0], alt[0], limbs, start)
        ctx.multiply_rns_poly(a[1], b[1], alt[2], limbs, start)
        ctx.multiply_rns_poly(a[0], b[1], alt[1], limbs, start)
        ctx.multiply_and_add_rns_poly(a[1], b[0], alt[1], alt[1], limbs, start)
        assert torch.equal(res, alt)
        return res

    e1q, e1b = behz_mul_1(ct1)
    e2q, e2b = behz_mul_1(ct2)
    pq = tensor(e1q, e2q, size_q, 0)
    pb = tensor(e1b, e2b, sk, aux0)
    t_q = np.array([plain_t % q for q in qp], dtype=np.uint64)
    t_q_s = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(t_q, qp)], dtype=np.uint64)
    t_b = np.array([plain_t % q for q in behz.bsk], dtype=np.uint64)
    t_b_s = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(t_b, behz.bsk)], dtype=np.uint64)
    out = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    tmp = zeros(sk)
    for i in range(3):   # :518-546
        ctx.nwt_2d_radix8_backward_inplace_scale(pq[i], size_q, 0, P.to_device(t_q, gpu), P.to_device(t_q_s, gpu))
        ctx.nwt_2d_radix8_backward_inplace_include_temp_mod_scale(pb[i], sk, 0, sk + 1, P.to_device(t_b, gpu), P.to_device(t_b_s, gpu))
        ctx.fast_floor(size_q, pq[i], pb[i], tmp)
        ctx.fastbconv_sk(size_q, tmp, out[i])
    whole = P.to_device(np.zeros((3, size_q, n), dtype=np.uint64), gpu)
    ctx.bfv_multiply_behz(P.to_device(ct1, gpu), P.to_device(ct2, gpu), whole)
    assert np.array_equal(P.to_host(out), P.to_host(whole))
    assert np.array_equal(P.to_host(out), behz.multiply(ct1, ct2))


@pytest.mark.parametrize("name,plain_t", [("c1_bfv4096", 65537), ("bfv13_50", 1032193), ("c4_bfv15", 1032193)])
def test_hps_scale_round_qr_r(name, plain_t, gpu):
    """scaleAndRound_HPS_QR_R (src/rns.cu:1700-1746) vs the oracle (same fma chain, alpha reduced in place across the R limbs)."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    hps = O.Hps(oc, plain_t)
    assert ctx.tool_aux_sizes(size_q)[1] == hps.size_r
    qr = list(primes[:size_q]) + list(hps.r)
    src = _extremes(uniform_poly(rng_for(2400), qr, n), qr)
    dst = P.to_device(np.zeros((hps.size_r, n), dtype=np.uint64), gpu)
    ctx.scaleAndRound_HPS_QR_R(size_q, dst, P.to_device(src, gpu))
    assert np.array_equal(P.to_host(dst), hps.scale_round_qr_r(src))
    # the tail of bfv_multiply_hps (src/evaluate.cu:800-808) from entries: scale-and-round, then R -> Q exactly (bConv_HPS)
    aux0 = _aux_row0(ctx, hps.r[0], len(primes))
    conv = P.DBaseConverter(ctx, list(range(aux0, aux0 + hps.size_r)), list(range(size_q)))
    back = P.to_device(np.zeros((size_q, n), dtype=np.uint64), gpu)
    conv.bConv_HPS(back, dst)
    assert np.array_equal(P.to_host(back), O.bconv_hps(hps.r, primes[:size_q], hps.scale_round_qr_r(src), n))


@pytest.mark.parametrize("name,plain_t,ql", [("bfv13_50", 65537, 4), ("bfv13_50", 65537, 2), ("hyb12_a2", 1032193, 4), ("c4_bfv15", 1032193, 30),
                                             ("c4_bfv15", 1032193, 15)])
def test_hps_overq_steps(name, plain_t, ql, gpu):
    """scaleAndRound_HPS_QlRl_Ql (src/rns.cu:1748-1796) at the top level and below it, ExpandCRTBasis_Ql_Q_add_to_ct (:1838-1858)
    where levels are dropped, vs the oracle."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ctx.set_plain_modulus(plain_t)
    hq = O.HpsOverQ(oc, plain_t, ql)
    assert ctx.tool_aux_sizes(ql)[2] == hq.size_r
    qr = list(primes[:ql]) + list(hq.r)
    r = rng_for(2500 + ql)
    src = _extremes(uniform_poly(r, qr, n), qr)
    dst = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.scaleAndRound_HPS_QlRl_Ql(ql, dst, P.to_device(src, gpu))
    assert np.array_equal(P.to_host(dst), hq.scale_round_qlrl_ql(src))
    low = _extremes(uniform_poly(r, primes[:ql], n), primes[:ql])
    acc = _extremes(uniform_poly(r, primes[:ql], n)[:, ::-1].copy(), primes[:ql])
    d_acc = P.to_device(acc, gpu)
    if ql < size_q:
        ctx.ExpandCRTBasis_Ql_Q_add_to_ct(ql, d_acc, P.to_device(low, gpu))
        assert np.array_equal(P.to_host(d_acc), hq.expand_add_to_ct(acc, low))
    else:
        with pytest.raises(ValueError, match="dropped"):
            ctx.ExpandCRTBasis_Ql_Q_add_to_ct(ql, d_acc, P.to_device(low, gpu))
