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


# (special-modulus size alpha, data limbs of the chain, live data limbs, batch): batches large enough for the product library to take
# modup_conv_s1_kernel (>= 1024 workgroups = 16 digit polynomials): both instantiations (15 inputs, <= 16 inputs zero-padded), short
# last digits (1, 2, 7 limbs), one digit only, the largest output count (45 + 15 - 1 = 59 limbs per digit)
FUSED_SWEEP = [(15, 45, 45, 6), (15, 45, 46 - 15, 8), (15, 45, 16, 8), (15, 45, 15, 16), (9, 27, 27, 6), (9, 27, 20, 6),
               (10, 30, 21, 6), (16, 32, 32, 8), (16, 32, 18, 8), (12, 24, 7, 16)]


@pytest.mark.parametrize("alpha,size_q,ql,batch", FUSED_SWEEP)
def test_batched_keyswitch_with_the_fused_conversion_equals_per_ciphertext_calls(alpha, size_q, ql, batch, gpu):
    """r05: at N = 2^16 a batched key switch whose mod-up launches >= 1024 workgroups converts inside the forward transform's strided pass
    (modup_conv_s1_kernel); one ciphertext at a time takes the separate conversion (inside the fused mod-up + inner product).  Both must
    give the same words for every ciphertext -- GPU against GPU, so the sweep over alpha / level / batch costs no oracle time (the
    oracle pins both paths at the C3 set and an alpha = 12 set in tests/test_gpu_rns.py)."""
    import torch
    import phantom_fhe_amd as P
    log_n, n = 16, 1 << 16
    primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * (size_q - 1) + [60] * alpha)]
    ctx = P.PhantomContext(log_n, primes, alpha, device=gpu)
    g = torch.Generator(device=gpu)
    g.manual_seed(0x5EED0500 + alpha * 64 + ql)
    rnd = lambda *s: torch.randint(0, 1 << 49, s, dtype=torch.int64, device=gpu, generator=g)   # below every prime of the set
    dnum = size_q // alpha
    rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(dnum)])
    ct, c2 = rnd(batch, 2, ql, n), rnd(batch, ql, n)
    want = ct.clone()
    for b in range(batch):
        ctx.keyswitch_inplace(ql, want[b], c2[b], rlk.public_keys_ptr, P.scheme_type.ckks)
    got = ct.clone()
    ctx.keyswitch_inplace_batched(ql, got, c2, batch, rlk.public_keys_ptr, P.scheme_type.ckks)
    assert torch.equal(got, want)
    if ql >= 2:      # ... and through the fused key switch + rescale
        out_b = torch.zeros((batch, 2, ql - 1, n), dtype=torch.int64, device=gpu)
        ctx.keyswitch_rescale_batched(ql, ct, c2, batch, rlk.public_keys_ptr, out_b)
        out_1 = torch.zeros_like(out_b)
        for b in range(batch):
            ctx.keyswitch_rescale(ql, ct[b], c2[b], rlk.public_keys_ptr, out_1[b])
        assert torch.equal(out_b, out_1)


def test_batched_keyswitch_rescale_with_the_fused_conversion_replays_from_a_hip_graph(gpu):
    """pha_keyswitch_rescale_batched at the C3 set with a batch that takes modup_conv_s1_kernel (76 KB of dynamic LDS: the launcher raises
    the kernel's limit on first use, which is why the header asks for one warm-up call before a capture): captured on an explicit stream
    after that warm-up, the graph replays on NEW inputs to what eager calls give."""
    import torch
    import phantom_fhe_amd as P
    log_n, n, alpha, size_q, ql, batch = 16, 1 << 16, 15, 45, 45, 6
    primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * (size_q - 1) + [60] * alpha)]
    ctx = P.PhantomContext(log_n, primes, alpha, device=gpu)
    g = torch.Generator(device=gpu)
    g.manual_seed(0x5EED0600)
    rnd = lambda *s: torch.randint(0, 1 << 49, s, dtype=torch.int64, device=gpu, generator=g)
    rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(size_q // alpha)])
    ins = [(rnd(batch, 2, ql, n), rnd(batch, ql, n)) for _ in range(3)]
    want = []
    for ct, c2 in ins:
        out = torch.zeros((batch, 2, ql - 1, n), dtype=torch.int64, device=gpu)
        ctx.keyswitch_rescale_batched(ql, ct, c2, batch, rlk.public_keys_ptr, out)
        want.append(out)
    d_ct, d_c2 = ins[0][0].clone(), ins[0][1].clone()
    out = torch.zeros((batch, 2, ql - 1, n), dtype=torch.int64, device=gpu)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ctx.keyswitch_rescale_batched(ql, d_ct, d_c2, batch, rlk.public_keys_ptr, out)    # warm-up on the capture stream
    side.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            ctx.keyswitch_rescale_batched(ql, d_ct, d_c2, batch, rlk.public_keys_ptr, out)
    for i in (1, 2, 0):
        d_ct.copy_(ins[i][0])
        d_c2.copy_(ins[i][1])
        out.zero_()
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want[i]), i
