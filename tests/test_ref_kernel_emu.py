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


# ---- r06: the reference's kernels WITH INDEX MAPS, emulated thread by thread, against the oracle's restatements ----
from util import oracle_ctx, primes_of, uniform_poly  # noqa: E402


def _tables(oc):
    """The DNTTTable rows of a context as the emulation wants them: [size_QP][n] arrays + per-row moduli / Barrett ratios / N^-1."""
    rows = range(oc.size_qp)
    tw = [oc.twiddle(i, 0) for i in rows]
    tws = [oc.twiddle(i, 1) for i in rows]
    itw = [oc.twiddle(i, 2) for i in rows]
    itws = [oc.twiddle(i, 3) for i in rows]
    mod = [int(q) for q in oc.primes]
    mu = [O.const_ratio(q) for q in mod]
    ninv = [oc.n_inv(i) for i in rows]
    ninvs = [O.compute_shoup(v, q) for v, q in zip(ninv, mod)]
    return tw, tws, itw, itws, mod, mu, ninv, ninvs


def _digit_constants(qlp, s, length):
    """QHatInvModq of the digit and QHatModp towards the complement of QlP, in Python integers (src/rns.cu:139-190,
    src/host/rns.cu:307-320, :448-457) -- computed here, NOT taken from the oracle."""
    part = [int(q) for q in qlp[s:s + length]]
    compl = [int(q) for j, q in enumerate(qlp) if j < s or j >= s + length]
    hat_inv, hats = [], []
    for i, qi in enumerate(part):
        hat = 1
        for k, qk in enumerate(part):
            if k != i:
                hat *= qk
        hats.append(hat)
        hat_inv.append(pow(hat % qi, -1, qi))
    mat = np.array([[h % p for h in hats] for p in compl], dtype=np.uint64)
    return part, compl, hat_inv, mat


@pytest.mark.parametrize("ql", [6, 5, 3])     # beta = 3 full digits; a ONE-limb last digit (2, 2, 1); two digits with a short one (2, 1)
@pytest.mark.parametrize("scheme", ["ckks", "bfv"])
def test_emulated_modup_kernels_equal_the_oracle(ql, scheme):
    """DRNSTool::modup (src/rns_bconv.cu:530-627) as the reference's kernels run it at N = 2^12, alpha = 2: inverse transform with the
    partQlHatInv scaling (ckks) or bconv_mult_kernel (bfv), modup_copy_partQl_kernel, bconv_matmul_padded_unroll2_kernel per digit
    (its thread map and the leap over the digit's own limbs), nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range (the
    twr_idx2 remap and the `continue`; bfv: the same launcher without a skipped range, fntt_2d.cu:407-618) == orc_modup."""
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_qp, size_q = len(primes), len(primes) - size_p
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql)
    tw, tws, itw, itws, mod, mu, ninv, ninvs = _tables(oc)
    alpha, qlp_n = size_p, ql + size_p
    beta = -(-ql // alpha)
    qlp = [int(p) for p in primes[:ql]] + [int(p) for p in primes[size_q:]]
    sch = O.CKKS if scheme == "ckks" else O.BFV
    cks = uniform_poly(rng_for(9300 + ql), primes[:ql], n)
    want = tool.modup(cks, sch)

    t_cks = cks.copy()
    if scheme == "ckks":   # nwt_2d_radix8_backward_scale(.., partQlHatInv ..) :558-559 = the inverse kernels + a full Shoup multiply per output
        for i in range(ql):
            s0 = (i // alpha) * alpha
            length = min(alpha, ql - s0)
            _, _, hat_inv, _ = _digit_constants(qlp, s0, length)
            hv = hat_inv[i - s0]
            x = E.nwt_2d_radix8_backward_inplace(cks[i], itw[i], itws[i], ninv[i], ninvs[i], mod[i])
            t_cks[i] = E._quiet(E._shoup)(x, np.uint64(hv), np.uint64(O.compute_shoup(hv, mod[i])), np.uint64(mod[i]))
    t_mod_up = np.zeros((beta, qlp_n, n), dtype=np.uint64)
    E.modup_copy_part_ql_kernel(t_mod_up, cks, ql, qlp_n, alpha, n)                       # :566-567
    for b in range(beta):
        s0 = alpha * b
        length = ql - alpha * (beta - 1) if b == beta - 1 else alpha                       # :572
        part, compl, hat_inv, mat = _digit_constants(qlp, s0, length)
        assert np.array_equal(t_mod_up[b, s0:s0 + length], cks[s0:s0 + length])           # the digit's own limbs, verbatim
        y = t_cks[s0:s0 + length]
        if scheme == "bfv":                                                                # :603-607
            y = E.bconv_mult_kernel(cks[s0:s0 + length], hat_inv, [O.compute_shoup(h, q) for h, q in zip(hat_inv, part)], part, n)
        E.bconv_matmul_padded_unroll2_kernel(t_mod_up[b], y, mat, compl, [O.const_ratio(p) for p in compl], length, n, s0, length)
        ex = (s0, s0 + length) if scheme == "ckks" else (0, 0)                             # :617-624
        visits = E.nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(t_mod_up[b], tw, tws, mod, n, qlp_n, 0, size_qp, size_p, *ex)
        rows = sorted({(v[1], v[2]) for v in visits})
        skipped = set(range(*ex))
        assert [r[0] for r in rows] == [j for j in range(qlp_n) if j not in skipped]
        assert all(r[1] == (r[0] if r[0] < ql else size_q + (r[0] - ql)) for r in rows)  # special limbs use the LAST size_P table rows
    assert np.array_equal(t_mod_up, want)


@pytest.mark.parametrize("ql", [6, 5, 1])
def test_emulated_moddown_kernels_equal_the_oracle(ql):
    """DRNSTool::moddown_from_NTT for ckks (src/rns_bconv.cu:776-828): nwt_2d_radix8_backward_inplace_include_special_mod on the P limbs,
    bConv_BEHZ (bconv_mult + bconv_matmul_unroll2_kernel), nwt_2d_radix8_forward_inplace_fuse_moddown with its epilogue
    (src/ntt/ntt_moddown.cu:203-208) == orc_moddown_from_ntt."""
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_qp, size_q = len(primes), len(primes) - size_p
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql)
    tw, tws, itw, itws, mod, mu, ninv, ninvs = _tables(oc)
    qlp = [int(p) for p in primes[:ql]] + [int(p) for p in primes[size_q:]]
    cx = uniform_poly(rng_for(9400 + ql), qlp, n)
    want = tool.moddown_from_ntt(cx, O.CKKS)
    work = cx.copy()
    E.nwt_2d_radix8_backward_inplace_include_special_mod(work, itw, itws, ninv, ninvs, mod, n, size_p, ql, size_qp, size_p)   # :790-793
    assert np.array_equal(work[:ql], cx[:ql])
    pp = [int(p) for p in primes[size_q:]]
    hat_inv, hats = [], []
    for i, pi in enumerate(pp):
        hat = 1
        for k, pk in enumerate(pp):
            if k != i:
                hat *= pk
        hats.append(hat)
        hat_inv.append(pow(hat % pi, -1, pi))
    y = E.bconv_mult_kernel(work[ql:], hat_inv, [O.compute_shoup(h, p) for h, p in zip(hat_inv, pp)], pp, n)
    out_primes = [int(p) for p in primes[:ql]]
    mat = np.array([[h % q for h in hats] for q in out_primes], dtype=np.uint64)
    delta = np.zeros((ql, n), dtype=np.uint64)
    E.bconv_matmul_padded_unroll2_kernel(delta, y, mat, out_primes, [O.const_ratio(q) for q in out_primes], size_p, n, ql, 0)   # no leap: :143-170
    big_p = 1
    for p in pp:
        big_p *= p
    pinv = [pow(big_p % q, -1, q) for q in out_primes]
    ct = np.zeros((ql, n), dtype=np.uint64)
    E.nwt_2d_radix8_forward_inplace_fuse_moddown(ct, work, pinv, [O.compute_shoup(v, q) for v, q in zip(pinv, out_primes)], delta, tw, tws, mod, n, ql, 0)
    assert np.array_equal(ct, want)


@pytest.mark.parametrize("ql", [6, 5, 2])
@pytest.mark.parametrize("threshold", [1 << 8, 0])
def test_emulated_inner_product_kernel_equals_the_oracle(ql, threshold):
    """key_switch_inner_prod_c2_and_evk (src/eval_key_switch.cu:14-69): keys at full QP width read through twr = nid >= size_Ql ? size_Q +
    (nid - size_Ql) : nid, data at QlP width; with the threshold callers pass (the mid-loop reduction never fires) and with 0 (it fires
    every term): the same canonical words either way, == orc_key_switch_inner_prod."""
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_qp, size_q = len(primes), len(primes) - size_p
    oc = oracle_ctx(name)
    tool = O.Tool(oc, ql)
    _, _, _, _, mod, mu, _, _ = _tables(oc)
    qlp = [int(p) for p in primes[:ql]] + [int(p) for p in primes[size_q:]]
    r = rng_for(9500 + ql)
    c2 = np.stack([uniform_poly(r, qlp, n) for _ in range(tool.beta)])
    evks = [np.stack([uniform_poly(r, primes, n), uniform_poly(r, primes, n)]) for _ in range(tool.beta)]
    want = tool.key_switch_inner_prod(c2, evks)
    got = E.key_switch_inner_prod_c2_and_evk(c2, evks, mod, mu, n, size_qp, ql + size_p, size_q, ql, tool.beta, threshold)
    assert np.array_equal(got.reshape(2, ql + size_p, n), want)


def test_emulated_barrett128_is_exact():
    """The PTX sequence of barrett_reduce_uint128_uint64 (uintmodmath.cuh:108-126) on random 128-bit values below q * 2^64 and on the
    largest accumulations the kernels form (16 products of 61-bit operands), against Python integers."""
    r = rng_for(9600)
    for bits in (36, 50, 60, 61):
        q = int(O.get_primes(4096, bits, 1)[0])
        mu0, mu1 = O.const_ratio(q)
        lo = r.integers(0, 1 << 63, 4096, dtype=np.uint64) * np.uint64(2) + r.integers(0, 2, 4096, dtype=np.uint64)
        hi = r.integers(0, q, 4096, dtype=np.uint64)
        hi[:8] = q - 1
        lo[:8] = np.uint64((1 << 64) - 1)
        got = E._quiet(E._barrett128)(lo, hi, np.uint64(q), np.uint64(mu0), np.uint64(mu1))
        for k in range(0, 4096, 37):
            assert int(got[k]) == ((int(hi[k]) << 64) | int(lo[k])) % q
