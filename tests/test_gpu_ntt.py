"""GPU parity: HIP NTT launchers (through the C ABI) vs the CPU oracle, bit-exact.

Reference analogue: test/ntt_test.cu:71-122 (round trip) -- here the forward values themselves are
pinned against the oracle, plus the variants used by key switching (include/ntt.cuh:178-226)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from util import oracle_ctx, primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu

# The product library has ONE plan per launch shape.  Every other geometry that was built and measured lives in the test-only
# experiments library (csrc/pha_experiments.h: pha_set_tuning); tests/test_gpu_ntt_variants.py re-runs this file in a process
# that loads it (PHA_LIB_OVERRIDE), which turns the variant sweep below on.
EXPERIMENTS = os.path.basename(os.environ.get("PHA_LIB_OVERRIDE", "")) == "libphantom_amd_exp.so"
needs_experiments = pytest.mark.skipif(not EXPERIMENTS, reason="needs libphantom_amd_exp.so (run by test_gpu_ntt_variants.py)")
DEFAULT_VARIANT = 1 | 32 | 64 | 2048 | 4096


def _ctx(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    return P.PhantomContext(log_n, list(primes), size_p, device=gpu)


@pytest.mark.parametrize("name", ["c1_bfv4096", "hyb13_a3", "c2_ntt14", "c4_bfv15", "c3_ckks16"])
def test_forward_inverse_inplace(name, gpu, ntt_variant):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc = oracle_ctx(name)
    ctx = _ctx(name, gpu)
    L = len(primes)
    x = uniform_poly(rng_for(1), primes, n)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace(d, L, 0)
    got = P.to_host(d)
    ref = oc.nwt_forward(x, L, 0)
    assert np.array_equal(got, ref)
    ctx.nwt_2d_radix8_backward_inplace(d, L, 0)
    assert np.array_equal(P.to_host(d), x)          # round trip (ntt_test.cu:100-110)
    # inverse values too (input need not be a forward image)
    d2 = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_backward_inplace(d2, L, 0)
    assert np.array_equal(P.to_host(d2), oc.nwt_backward(x, L, 0))


# 353 / 361: bit 8 forces the one-launch plans of N = 8192 and N = 16384 for every launch size
# 609 / 617 / 625: bit 9 sends every launch through the one-launch form (both passes in one kernel, L2 hand-off)
# 2145: the default (bit 11: polynomial-fastest block order in batched contiguous passes)
# 6241: the default (bit 12: N = 2^16 as 64 x 1024); 10337 / 18529: the one-wavefront 1024-point rows, 128 x 512
_VARIANTS = [0, 1, 8, 9, 17, 25, 65, 73, 81, 225, 353, 361, 609, 617, 625, 1121, 2145, 6241, 10337, 18529]
_VARIANT_IDS = ["ept16", "ept8", "ept16-int", "ept8-int", "ept8-ot", "ept8-ot-int", "ept8-wave", "ept8-wave-int", "ept8-ot-wave",
                "two-pass-4096", "one-launch-8192-16384", "one-launch-8192-16384-int", "fused", "fused-int", "fused-ot", "never-fused",
                "r03-default", "default", "rows1024-one-wavefront", "split-128x512"]


@pytest.fixture(params=_VARIANTS if EXPERIMENTS else [DEFAULT_VARIANT], ids=_VARIANT_IDS if EXPERIMENTS else ["product"])
def ntt_variant(request):
    """Every thread geometry of the NTT (experiments library: pha_set_tuning key 0) must give identical results; the product
    library runs its one plan."""
    import phantom_fhe_amd as P
    if not EXPERIMENTS:
        assert not P.has_tuning()      # the product library does not export the knob
        yield request.param
        return
    P.set_tuning(0, request.param)
    yield request.param
    P.set_tuning(0, DEFAULT_VARIANT)


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
def test_all_degrees(log_n, gpu, ntt_variant):
    import phantom_fhe_amd as P
    n = 1 << log_n
    primes = [int(p) for p in O.coeff_modulus_create(n, [60, 50, 40])]
    oc = O.Ctx(log_n, primes, 0)
    ctx = P.PhantomContext(log_n, primes, 0, device=gpu)
    x = uniform_poly(rng_for(100 + log_n), primes, n)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace(d, 3, 0)
    assert np.array_equal(P.to_host(d), oc.nwt_forward(x, 3, 0))
    ctx.nwt_2d_radix8_backward_inplace(d, 3, 0)
    assert np.array_equal(P.to_host(d), x)
    # tables equal the oracle's (DNTTTable content, include/ntt.cuh:34-129), incl. folded itwiddle[1]
    for which in range(4):
        assert np.array_equal(ctx.twiddle_row(1, which), oc.twiddle(1, which))
    info = ctx.prime_info(0)
    assert info["value"] == primes[0] and info["const_ratio"] == O.const_ratio(primes[0])
    assert info["root"] == O.minimal_primitive_root(2 * n, primes[0]) and info["n_inv"] == oc.n_inv(0)


def test_start_index_and_partial(gpu):
    """start_modulus_idx / coeff_modulus_size select limbs [start, start+size) (fntt_2d.cu:40-41)."""
    import phantom_fhe_amd as P
    name = "hyb13_a3"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    x = uniform_poly(rng_for(2), primes, n)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace(d, 3, 2)
    ref = x.copy()
    ref[2:5] = oc.nwt_forward(x[2:5], 3, 2)
    assert np.array_equal(P.to_host(d), ref)
    ctx.nwt_2d_radix8_backward_inplace(d, 1, 4)     # rescale uses size 1 at the last limb (rns.cu:1171)
    ref[4:5] = oc.nwt_backward(ref[4:5], 1, 4)
    assert np.array_equal(P.to_host(d), ref)


def test_special_mod_and_exclude_range(gpu, ntt_variant):
    """[Ql || P] buffers: P limbs use the last rows of the table (fntt_2d.cu:434-437); the digit's own
    range is skipped (ntt_modup.cu:422)."""
    import phantom_fhe_amd as P
    name = "hyb13_a3"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_qp, size_q = len(primes), len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    ql = 5
    qlp_idx = list(range(ql)) + [size_q + i for i in range(size_p)]
    x = uniform_poly(rng_for(3), [primes[i] for i in qlp_idx], n)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod(d, ql + size_p, 0, size_qp, size_p)
    ref = oc.nwt_forward_map(x, qlp_idx)
    assert np.array_equal(P.to_host(d), ref)
    ctx.nwt_2d_radix8_backward_inplace_include_special_mod(d, size_p, ql, size_qp, size_p)   # moddown's call
    ref2 = ref.copy()
    ref2[ql:] = oc.nwt_backward_map(ref[ql:], qlp_idx[ql:])
    assert np.array_equal(P.to_host(d), ref2)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(d, ql + size_p, 0, size_qp, size_p, 3, 5)
    ref3 = oc.nwt_forward_map(x, qlp_idx)
    ref3[3:5] = x[3:5]
    assert np.array_equal(P.to_host(d), ref3)


def test_backward_out_of_place_and_scale(gpu, ntt_variant):
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = 6
    x = uniform_poly(rng_for(4), primes[:L], n)
    d_in = P.to_device(x, gpu)
    d_out = P.to_device(np.zeros_like(x), gpu)
    ctx.nwt_2d_radix8_backward(d_out, d_in, L, 0)
    ref = oc.nwt_backward(x, L, 0)
    assert np.array_equal(P.to_host(d_out), ref)
    assert np.array_equal(P.to_host(d_in), x)       # source untouched
    scale = np.array([rng_for(5).integers(1, int(q)) for q in primes[:L]], dtype=np.uint64)
    shoup = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(scale, primes[:L])], dtype=np.uint64)
    ctx.nwt_2d_radix8_backward_scale(d_out, d_in, L, 0, P.to_device(scale, gpu), P.to_device(shoup, gpu))
    assert np.array_equal(P.to_host(d_out), oc.multiply_scalar(ref, scale, L, 0))


def test_forward_fuse_moddown(gpu, ntt_variant):
    """ct = (cx - NTT(delta)) * PInv (ntt_moddown.cu:203-208), also with ct aliasing cx."""
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = 5
    r = rng_for(6)
    cx = uniform_poly(r, primes[:L], n)
    delta = uniform_poly(r, primes[:L], n)
    c = np.array([r.integers(1, int(q)) for q in primes[:L]], dtype=np.uint64)
    cs = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(c, primes[:L])], dtype=np.uint64)
    ref = oc.multiply_scalar(oc.sub(cx, oc.nwt_forward(delta, L, 0), L, 0), c, L, 0)
    d_cx, d_delta = P.to_device(cx, gpu), P.to_device(delta, gpu)
    ctx.nwt_2d_radix8_forward_inplace_fuse_moddown(d_cx, d_cx, P.to_device(c, gpu), P.to_device(cs, gpu), d_delta, L, 0)
    assert np.array_equal(P.to_host(d_cx), ref)


def test_invalid_arguments(gpu):
    import phantom_fhe_amd as P
    ctx = _ctx("c1_bfv4096", gpu)
    d = P.to_device(np.zeros((3, 4096), dtype=np.uint64), gpu)
    with pytest.raises(ValueError):
        ctx.nwt_2d_radix8_forward_inplace(d, 4, 0)   # beyond the table
    with pytest.raises(ValueError):
        P.PhantomContext(12, [97, 193], 0, device=gpu)  # not NTT primes for N=4096
    with pytest.raises(ValueError):
        P.PhantomContext(9, [0xffffee001], 0, device=gpu)


def test_empty_and_full_size_properties(gpu):
    """Edge cases and size-independent properties at the largest supported shape (N = 2^17, 60 limbs):
    zero limbs is a no-op; inverse(forward(x)) = x; forward is linear; forward of 0 is 0."""
    import phantom_fhe_amd as P
    log_n, n = 17, 1 << 17
    primes = [int(p) for p in P.coeff_modulus_create(n, [60] * 16 + [40] * 44)]
    ctx = P.PhantomContext(log_n, primes, 15, device=gpu)
    r = rng_for(97)
    a = uniform_poly(r, primes, n)
    b = uniform_poly(r, primes, n)
    b[:, :64] = np.array(primes, dtype=np.uint64)[:, None] - 1          # q - 1 runs
    da, db = P.to_device(a, gpu), P.to_device(b, gpu)
    ctx.nwt_2d_radix8_forward_inplace(da, 0, 0)                           # empty limb range
    assert np.array_equal(P.to_host(da), a)
    dsum = P.to_device(np.zeros_like(a), gpu)
    ctx.add_rns_poly(da, db, dsum, 60)
    for t in (da, db, dsum):
        ctx.nwt_2d_radix8_forward_inplace(t, 60, 0)
    lin = P.to_device(np.zeros_like(a), gpu)
    ctx.add_rns_poly(da, db, lin, 60)
    assert np.array_equal(P.to_host(lin), P.to_host(dsum))               # NTT(a) + NTT(b) == NTT(a + b)
    ha = P.to_host(da)
    assert all(int(ha[l].max()) < primes[l] for l in range(60))          # canonical outputs
    ctx.nwt_2d_radix8_backward_inplace(da, 60, 0)
    assert np.array_equal(P.to_host(da), a)
    dz = P.to_device(np.zeros_like(a), gpu)
    ctx.nwt_2d_radix8_forward_inplace(dz, 60, 0)
    assert not P.to_host(dz).any()


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
def test_large_residues_fp_thresholds(log_n, gpu, ntt_variant):
    """Residues just below q everywhere (busy low bits): the unreduced FP64 sums of the light butterflies reach
    their worst case, right at the prime sizes where the light / deferred variants switch."""
    import phantom_fhe_amd as P
    n = 1 << log_n
    primes = [int(p) for p in O.coeff_modulus_create(n, [50, 47, 46, 42, 41, 40, 39, 30])]
    oc = O.Ctx(log_n, primes, 0)
    ctx = P.PhantomContext(log_n, primes, 0, device=gpu)
    r = rng_for(700 + log_n)
    x = np.stack([(q - 1 - r.integers(0, 1 << 16, n)).astype(np.uint64) for q in primes])
    L = len(primes)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_backward_inplace(d, L, 0)
    assert np.array_equal(P.to_host(d), oc.nwt_backward(x, L, 0))
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace(d, L, 0)
    assert np.array_equal(P.to_host(d), oc.nwt_forward(x, L, 0))


def test_temp_mod_variants_transform_the_behz_base(gpu):
    """include_temp_mod / include_temp_mod_scale (fntt_2d.cu:200-405, intt_2d.cu:313-409): NTTs over Bsk = B u {m_sk}."""
    import phantom_fhe_amd as P
    name = "hyb12_a2"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    d = P.to_device(np.zeros((4, n), dtype=np.uint64), gpu)
    with pytest.raises(ValueError):
        ctx.nwt_2d_radix8_forward_inplace_include_temp_mod(d, 4, 0, 5)       # no plain modulus yet: no BEHZ base
    ctx.set_plain_modulus(65537)
    behz = O.Behz(oc, 65537)
    sk = behz.size_bsk
    ob = O.Ctx(log_n, behz.bsk, 0)
    x = uniform_poly(rng_for(41), behz.bsk, n)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_include_temp_mod(d, sk, 0, sk + 1)
    fwd = ob.nwt_forward(x, sk, 0)
    assert np.array_equal(P.to_host(d), fwd)
    scale = np.array([65537 % q for q in behz.bsk], dtype=np.uint64)        # tModBsk (evaluate.cu:530)
    shoup = np.array([O.compute_shoup(int(s), int(q)) for s, q in zip(scale, behz.bsk)], dtype=np.uint64)
    ctx.nwt_2d_radix8_backward_inplace_include_temp_mod_scale(d, sk, 0, sk + 1, P.to_device(scale, gpu), P.to_device(shoup, gpu))
    assert np.array_equal(P.to_host(d), ob.multiply_scalar(x, scale, sk, 0))
    with pytest.raises(ValueError):
        ctx.nwt_2d_radix8_forward_inplace_include_temp_mod(d, sk - 1, 0, sk + 1)


@pytest.mark.parametrize("name", ["hyb12_a2", "hyb13_a3", "c2_ntt14"])
def test_forward_modup_fuse(name, gpu, ntt_variant):
    """nwt_2d_radix8_forward_modup_fuse (ntt_keyswitch_old.cu:225-265): a plaintext modulo t lifted into one RNS limb
    at a time, the way add_plain / multiply_plain call it for BGV (evaluate.cu:1150-1154)."""
    import phantom_fhe_amd as P
    log_n, primes, _ = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = min(len(primes), 5)
    t = 65537
    plain = rng_for(43).integers(0, t, (1, n), dtype=np.uint64)
    d_plain = P.to_device(plain, gpu)
    d_out = P.to_device(np.zeros((L, n), dtype=np.uint64), gpu)
    for i in range(L):
        ctx.nwt_2d_radix8_forward_modup_fuse(d_out[i:i + 1], d_plain, i, 1, 0)
    want = oc.nwt_forward(np.repeat(plain, L, axis=0), L, 0)
    assert np.array_equal(P.to_host(d_out), want)
    assert np.array_equal(P.to_host(d_plain), plain)
    # several limbs at once: limbs [1, 3) of `in`, all modulo prime 2; the rest of `out` is left alone
    src = uniform_poly(rng_for(44), [primes[2]] * 4, n)
    d_o = P.to_device(np.full((4, n), 7, dtype=np.uint64), gpu)
    ctx.nwt_2d_radix8_forward_modup_fuse(d_o, P.to_device(src, gpu), 2, 2, 1)
    got = P.to_host(d_o)
    assert np.array_equal(got[1:3], oc.nwt_forward_map(src[1:3], [2, 2]))
    assert (got[0] == 7).all() and (got[3] == 7).all()


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 10, 11])
def test_single_workgroup_transforms(log_n, gpu):
    """fnwt_1d(_opt) / inwt_1d(_opt) (src/ntt/ntt_1d.cu; test/ntt_test.cu:9-69 is the reference's own use): caller-built
    tables, N <= 2048, round trip and values vs the oracle; the inverse scales only the first half of its outputs."""
    import phantom_fhe_amd as P
    n = 1 << log_n
    primes = [int(p) for p in O.get_primes(max(n, 2), 50, 2)] + [int(O.get_primes(max(n, 2), 60, 1)[0])]
    L = len(primes)
    tabs = [O.ntt_tables(log_n, q) for q in primes]
    tw = np.stack([t[0] for t in tabs]); tws = np.stack([t[1] for t in tabs])
    itw = np.stack([t[2] for t in tabs]); itws = np.stack([t[3] for t in tabs])
    ninv = np.array([t[4] for t in tabs], dtype=np.uint64); ninvs = np.array([t[5] for t in tabs], dtype=np.uint64)
    mod = np.array([[q, O.const_ratio(q)[0], O.const_ratio(q)[1]] for q in primes], dtype=np.uint64)
    x = uniform_poly(rng_for(900 + log_n), primes, n)
    x[0, : n // 2] = primes[0] - 1
    oc = O.Ctx(log_n, primes, 0)
    want = oc.nwt_forward(x, L, 0)
    dev = lambda a: P.to_device(np.ascontiguousarray(a), gpu)
    d_tw, d_tws, d_itw, d_itws, d_mod = dev(tw), dev(tws), dev(itw), dev(itws), dev(mod)
    for opt in ((False, True) if n >= 4 else (False,)):
        d = dev(x)
        P.fnwt_1d(d, d_tw, d_tws, d_mod, n, L, 0, opt=opt)
        assert np.array_equal(P.to_host(d), want)
        # the oracle's inverse table already has N^-1 folded into slot 1 (src/host/ntt.cu:53-55), so scalar = N^-1
        # completes the inverse
        P.inwt_1d(d, d_itw, d_itws, d_mod, dev(ninv), dev(ninvs), n, L, 0, opt=opt)
        assert np.array_equal(P.to_host(d), x)
    # a limb range, and the quirk of the _opt forward launcher (it ignores start_modulus_idx)
    d = dev(x)
    P.fnwt_1d(d, d_tw, d_tws, d_mod, n, 1, 2)
    ref = x.copy(); ref[2] = want[2]
    assert np.array_equal(P.to_host(d), ref)
    if n >= 4:
        d = dev(x)
        P.fnwt_1d(d, d_tw, d_tws, d_mod, n, 1, 2, opt=True)
        ref = x.copy(); ref[0] = want[0]
        assert np.array_equal(P.to_host(d), ref)
    # scalar = 1: the first half stays unscaled, i.e. N times the true inverse there
    one = np.ones(L, dtype=np.uint64)
    one_s = np.array([O.compute_shoup(1, q) for q in primes], dtype=np.uint64)
    d = dev(want)
    P.inwt_1d(d, d_itw, d_itws, d_mod, dev(one), dev(one_s), n, L, 0)
    got = P.to_host(d)
    nvec = np.array([n % q for q in primes], dtype=np.uint64)
    assert np.array_equal(got[:, n // 2:], x[:, n // 2:])
    assert np.array_equal(got[:, : n // 2], oc.multiply_scalar(x, nvec, L, 0)[:, : n // 2])
    with pytest.raises(ValueError):
        P.fnwt_1d(dev(np.zeros((1, 4096), dtype=np.uint64)), d_tw, d_tws, d_mod, 4096, 1, 0)


def test_one_launch_8192_takes_over_for_large_launches(gpu):
    """N = 8192: launches of 64 or more limb-polynomials take the one-launch plan by default (smaller ones keep the two
    passes); both must agree with the oracle, also with the special-prime remap and a skipped digit range."""
    import phantom_fhe_amd as P
    name = "hyb13_a3"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L, batch = len(primes), 8                                     # 12 x 8 = 96 limb-polynomials
    x = np.stack([uniform_poly(rng_for(50 + b), primes, n) for b in range(batch)])
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_batched(d, L, 0, batch, L * n)
    want = np.stack([oc.nwt_forward(x[b], L, 0) for b in range(batch)])
    assert np.array_equal(P.to_host(d), want)
    ctx.nwt_2d_radix8_backward_inplace_batched(d, L, 0, batch, L * n)
    assert np.array_equal(P.to_host(d), x)


@pytest.mark.parametrize("batch", [8, 9, 16])
def test_batched_forward_with_resident_twiddles_at_2_16(batch, gpu):
    """r04: plain forward launches of >= 8 polynomials at N = 2^16 take the twiddle-resident contiguous pass (ntt_cpass_zloop_kernel:
    the FP64 limbs walk `zper` polynomials per workgroup, the integer limbs run one (tile, polynomial) per workgroup at the head of
    the same grid).  The C3 chain mixes both back ends (limb 0 is a 60-bit prime, limbs 1..44 are 50-bit); every polynomial against
    the oracle, odd batch sizes included, plus a limb range that starts inside the chain."""
    import phantom_fhe_amd as P
    name = "c3_ckks16"
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = len(primes) - size_p
    x = np.stack([uniform_poly(rng_for(700 + z), primes[:L], n) for z in range(batch)])
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_batched(d, L, 0, batch, L * n)
    got = P.to_host(d)
    for z in range(batch):
        assert np.array_equal(got[z], oc.nwt_forward(x[z], L, 0)), z
    ctx.nwt_2d_radix8_backward_inplace_batched(d, L, 0, batch, L * n)
    assert np.array_equal(P.to_host(d), x)
    # limbs [0, 7) only (one integer limb + six FP64 limbs), the other limbs must stay untouched
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_batched(d, 7, 0, batch, L * n)
    got = P.to_host(d)
    for z in range(0, batch, 3):
        assert np.array_equal(got[z, :7], oc.nwt_forward(x[z, :7], 7, 0)) and np.array_equal(got[z, 7:], x[z, 7:])


@needs_experiments
@pytest.mark.parametrize("lag", [0, 1, 2, 5])
@pytest.mark.parametrize("name,batch", [("c2_ntt14", 3), ("c4_bfv15", 2), ("c3_ckks16", 2)])
def test_one_launch_transform_lags_and_batches(name, batch, lag, gpu):
    """Both passes in one launch (the intermediate handed over through the XCD's L2) for every lag between the
    passes, batches of polynomials, forward and inverse: bit-identical to the oracle, and the hand-off counters are
    clean again afterwards (a second call gives the same answer)."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = len(primes)
    x = np.stack([uniform_poly(rng_for(300 + z), primes, n) for z in range(batch)])
    P.set_tuning(0, 1 | 32 | 64 | 512)
    P.set_tuning(3, lag)
    try:
        for rep in range(2):
            d = P.to_device(x, gpu)
            ctx.nwt_2d_radix8_forward_inplace_batched(d, L, 0, batch, L * n)
            got = P.to_host(d)
            for z in range(batch):
                assert np.array_equal(got[z], oc.nwt_forward(x[z], L, 0)), (rep, z)
            ctx.nwt_2d_radix8_backward_inplace_batched(d, L, 0, batch, L * n)
            assert np.array_equal(P.to_host(d), x)
    finally:
        P.set_tuning(0, DEFAULT_VARIANT)
        P.set_tuning(3, 2)


@needs_experiments
@pytest.mark.parametrize("name,batch", [("c2_ntt14", 5), ("c4_bfv15", 3), ("c3_ckks16", 4)])
def test_batched_launches_in_both_block_orders(name, batch, gpu):
    """The polynomial-fastest, XCD-grouped block order of batched contiguous passes (default) against the plain 3-D grid:
    identical results, forward and inverse, batch sizes on both sides of the on-the-fly-twiddle rule."""
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    L = len(primes)
    x = np.stack([uniform_poly(rng_for(400 + z), primes, n) for z in range(batch)])
    outs = []
    try:
        for variant in (1 | 32 | 64, 1 | 32 | 64 | 2048):
            P.set_tuning(0, variant)
            d = P.to_device(x, gpu)
            ctx.nwt_2d_radix8_forward_inplace_batched(d, L, 0, batch, L * n)
            outs.append(P.to_host(d))
            ctx.nwt_2d_radix8_backward_inplace_batched(d, L, 0, batch, L * n)
            assert np.array_equal(P.to_host(d), x)
    finally:
        P.set_tuning(0, DEFAULT_VARIANT)
    assert np.array_equal(outs[0], outs[1])
    for z in range(batch):
        assert np.array_equal(outs[1][z], oc.nwt_forward(x[z], L, 0))


def test_batched_transforms_replay_from_a_hip_graph(gpu):
    """The batched twiddle-resident pass (r04) is a plain kernel launch with its work map passed by value: a forward + inverse batched
    transform at N = 2^16 (8 polynomials x 45 limbs: integer and FP64 limbs in the one grid) captured into a hipGraph replays to
    the eager result, twice."""
    import torch
    import phantom_fhe_amd as P
    name, batch = "c3_ckks16", 8
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    ctx = _ctx(name, gpu)
    L = len(primes) - size_p
    x = np.stack([uniform_poly(rng_for(900 + z), primes[:L], n) for z in range(batch)])
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace_batched(d, L, 0, batch, L * n)
    want = P.to_host(d)
    ctx.nwt_2d_radix8_backward_inplace_batched(d, L, 0, batch, L * n)
    assert np.array_equal(P.to_host(d), x)
    side = torch.cuda.Stream(device=gpu)
    fwd = P.to_device(x, gpu)
    with torch.cuda.stream(side):
        ctx.nwt_2d_radix8_forward_inplace_batched(fwd, L, 0, batch, L * n)     # warm-up on the capture stream
        fwd.copy_(P.to_device(x, gpu))
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ctx.nwt_2d_radix8_forward_inplace_batched(fwd, L, 0, batch, L * n)
    for rep in range(2):
        fwd.copy_(P.to_device(x, gpu))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(P.to_host(fwd), want), rep
