"""BASELINE configs 4 and 5 as host compositions (phantom_fhe_amd/workloads.py) on the GPU vs the oracle's
composition of the same reference steps, and: the union of the rank shards is bit-identical to the unsharded run
for every world size ("identical results on 1/2/4/8 GPUs")."""
import numpy as np
import pytest

from oracle import oracle as O
from util import oracle_ctx, primes_of, rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _ctx(name, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p = primes_of(name)
    return P.PhantomContext(log_n, list(primes), size_p, device=gpu)


def _keys(rng, primes, n, dnum):
    return np.stack([np.stack([uniform_poly(rng, primes, n), uniform_poly(rng, primes, n)]) for _ in range(dnum)])


@pytest.mark.parametrize("name,scheme,ql,batch", [("hyb12_a2", O.BFV, 6, 5), ("hyb12_a2", O.CKKS, 4, 3), ("c4_bfv15", O.BFV, 30, 2),
                                                  ("hyb13_a3", O.CKKS, 7, 3), ("hyb13_a3", O.BFV, 9, 2), ("c1_bfv4096", O.BFV, 2, 4),
                                                  ("c1_bfv4096", O.CKKS, 2, 1), ("hyb12_a2", O.BGV, 6, 2)])
def test_config4_relinearize_rotate_batch(name, scheme, ql, batch, gpu):
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        ctx.set_plain_modulus(65537)
        tool.set_plain_modulus(65537)
    r = rng_for(800 + batch)
    rlk, glk = _keys(r, primes, n, size_q // size_p), _keys(r, primes, n, size_q // size_p)
    elt = 3
    ct3 = np.stack([np.stack([uniform_poly(r, primes[:ql], n) for _ in range(3)]) for _ in range(batch)])
    d_rlk, d_glk = P.PhantomRelinKey.from_numpy(rlk, gpu), P.PhantomRelinKey.from_numpy(glk, gpu)
    d_ct3 = P.to_device(ct3, gpu)
    full = P.to_host(W.relinearize_rotate_batch(ctx, ql, d_ct3, d_rlk, d_glk, elt, scheme))      # pha_relinearize_rotate_batched
    assert np.array_equal(P.to_host(d_ct3), ct3)                                                  # the input is only read
    for chunk in (1, 2, 0):
        assert np.array_equal(P.to_host(W.relinearize_rotate_batch(ctx, ql, d_ct3, d_rlk, d_glk, elt, scheme, chunk=chunk)), full)
    assert np.array_equal(P.to_host(W.relinearize_rotate_batch_host(ctx, ql, d_ct3, d_rlk, d_glk, elt, scheme)), full)
    # the oracle's composition of the same reference steps, ciphertext by ciphertext
    table = O.galois_ntt_table(log_n, elt)
    for b in range(batch):
        ct = tool.keyswitch_inplace(ct3[b, :2], ct3[b, 2], [rlk[i] for i in range(tool.beta)], scheme)
        if scheme == O.BFV:
            g = [oc.apply_galois_coeff(ct[p], elt, ql) for p in range(2)]
        else:
            g = [O.apply_galois_ntt(ct[p], table, n, ql) for p in range(2)]
        want = tool.keyswitch_inplace(np.stack([g[0], np.zeros_like(g[0])]), g[1], [glk[i] for i in range(tool.beta)], scheme)
        assert np.array_equal(full[b], want)
    for world in (2, 4, 8):
        parts, covered = [], []
        for rank in range(world):
            mine, res = W.relinearize_rotate_sharded(ctx, ql, d_ct3, d_rlk, d_glk, elt, scheme, rank=rank, world=world)
            covered += list(mine)
            parts.append(P.to_host(res))
        assert covered == list(range(batch))
        assert np.array_equal(np.concatenate(parts), full)


@pytest.mark.parametrize("name,ql,n_blocks,n_diag", [("hyb12_a2", 5, 3, 4), ("hyb13_a3", 9, 2, 16)])
def test_config5_diagonal_matvec_row_blocks(name, ql, n_blocks, n_diag, gpu):
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(900 + n_diag)
    elts = [1] + [int(pow(5, k, 2 * n)) for k in range(1, n_diag)]          # rotations by 0 .. n_diag-1 slots
    glk = [None] + [_keys(r, primes, n, size_q // size_p) for _ in elts[1:]]
    qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
    blocks = [[uniform_poly(r, qlp_primes, n) for _ in elts] for _ in range(n_blocks)]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_keys = [None] + [P.PhantomRelinKey.from_numpy(k, gpu) for k in glk[1:]]
    d_blocks = [[P.to_device(w, gpu) for w in blk] for blk in blocks]
    d_ct = P.to_device(ct, gpu)
    mine, outs = W.matvec_row_blocks_sharded(ctx, ql, d_ct, elts, d_keys, d_blocks, O.CKKS, rank=0, world=1)
    assert list(mine) == list(range(n_blocks))
    assert np.array_equal(P.to_host(d_ct), ct)                               # the input vector is not consumed
    full = [P.to_host(o) for o in outs]
    okeys = [None] + [[k[i] for i in range(tool.beta)] for k in glk[1:]]
    for i in range(n_blocks):
        assert np.array_equal(full[i], tool.hoisting_weighted(ct, elts, okeys, blocks[i], O.CKKS))
    for world in (2, 8):
        got = {}
        for rank in range(world):
            mine, outs = W.matvec_row_blocks_sharded(ctx, ql, d_ct, elts, d_keys, d_blocks, O.CKKS, rank=rank, world=world)
            for i, o in zip(mine, outs):
                got[i] = P.to_host(o)
        assert sorted(got) == list(range(n_blocks))
        assert all(np.array_equal(got[i], full[i]) for i in range(n_blocks))


def _gpu_uniform(primes, n, gpu, gen):
    import torch
    out = torch.empty((len(primes), n), dtype=torch.int64, device=gpu)
    for i, q in enumerate(primes):
        out[i] = torch.randint(0, int(q), (n,), dtype=torch.int64, device=gpu, generator=gen)
    return out


def test_config4_two_internal_streams_replay_from_a_hip_graph(gpu):
    """With its own chunking pha_relinearize_rotate_batched alternates the sets between two streams the context owns (forked from
    and joined back into the caller's stream by events): the result equals the one-stream form, and the whole call -- fork, both
    lanes, join -- can be captured into a hipGraph on a side stream (after one warm-up call) and replayed."""
    import torch
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, batch, elt = "c4_bfv15", 30, 16, 3          # 16 ciphertexts: two sets of four per lane at this shape
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ctx = _ctx(name, gpu)
    r = rng_for(4016)
    rlk, glk = _keys(r, primes, n, size_q // size_p), _keys(r, primes, n, size_q // size_p)
    d_rlk, d_glk = P.PhantomRelinKey.from_numpy(rlk, gpu), P.PhantomRelinKey.from_numpy(glk, gpu)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(0x5EED4016)
    ct3 = torch.stack([torch.stack([_gpu_uniform(primes[:ql], n, gpu, gen) for _ in range(3)]) for _ in range(batch)])
    one_stream = W.relinearize_rotate_batch(ctx, ql, ct3, d_rlk, d_glk, elt, O.BFV, chunk=8)
    assert torch.equal(W.relinearize_rotate_batch(ctx, ql, ct3, d_rlk, d_glk, elt, O.BFV), one_stream)    # chunk = 0: two lanes
    out = torch.empty_like(ct3[:, :2])
    side = torch.cuda.Stream(device=gpu)

    def call():
        ctx.relinearize_rotate_batched(ql, ct3, batch, d_rlk.public_keys_ptr, d_glk.public_keys_ptr, elt, O.BFV, out, 0)

    with torch.cuda.stream(side):
        call()                                            # warm-up: the lanes and their arenas exist from here on
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call()
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, one_stream)


def test_config4_two_internal_streams_from_concurrent_host_threads(gpu):
    """Two host threads, each on its own stream, go through the context's two internal streams at the same time (the enqueue is
    serialised inside the entry, the fork / join events are re-recorded per call): both get the one-stream result, every time."""
    import threading
    import torch
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, batch, elt = "c4_bfv15", 30, 16, 3
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    ctx = _ctx(name, gpu)
    r = rng_for(4017)
    rlk, glk = _keys(r, primes, n, size_q // size_p), _keys(r, primes, n, size_q // size_p)
    d_rlk, d_glk = P.PhantomRelinKey.from_numpy(rlk, gpu), P.PhantomRelinKey.from_numpy(glk, gpu)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(0x5EED4017)
    inputs = [torch.stack([torch.stack([_gpu_uniform(primes[:ql], n, gpu, gen) for _ in range(3)]) for _ in range(batch)]) for _ in range(2)]
    want = [W.relinearize_rotate_batch(ctx, ql, x, d_rlk, d_glk, elt, O.BFV, chunk=8) for x in inputs]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(gpu)
            st = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(st):
                for _ in range(3):
                    got = W.relinearize_rotate_batch(ctx, ql, inputs[i], d_rlk, d_glk, elt, O.BFV)
                    st.synchronize()
                    if not torch.equal(got, want[i]):
                        errors.append(f"thread {i}: result differs")
        except Exception as e:   # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_config4_batch_64_at_its_stated_shape(gpu):
    """BASELINE config 4 as written: BFV relinearize + Galois rotate at N = 2^15, 30 + 15 limbs, a batch of 64
    ciphertexts.  Four sampled ciphertexts are checked against the oracle's composition of the reference steps; the
    whole batch must give the same per-ciphertext digests for every chunking of the batched key switch and for every
    world size 1 / 2 / 4 / 8 ("identical results on 1/2/4/8 GPUs")."""
    import torch
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, batch, elt = "c4_bfv15", 30, 64, 3
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(4064)
    rlk, glk = _keys(r, primes, n, size_q // size_p), _keys(r, primes, n, size_q // size_p)
    d_rlk, d_glk = P.PhantomRelinKey.from_numpy(rlk, gpu), P.PhantomRelinKey.from_numpy(glk, gpu)
    gen = torch.Generator(device=gpu)
    gen.manual_seed(0x5EED4064)
    ct3 = torch.stack([torch.stack([_gpu_uniform(primes[:ql], n, gpu, gen) for _ in range(3)]) for _ in range(batch)])

    def digests(t):   # one wrapping 64-bit sum per ciphertext
        return [int(x) for x in t.reshape(t.shape[0], -1).sum(dim=1).cpu()]

    full = W.relinearize_rotate_batch(ctx, ql, ct3, d_rlk, d_glk, elt, O.BFV)            # chunks of 8 (the default)
    want = digests(full)
    for b in (0, 21, 42, 63):                                                             # oracle, ciphertext by ciphertext
        x = P.to_host(ct3[b])
        ct = tool.keyswitch_inplace(x[:2], x[2], [rlk[i] for i in range(tool.beta)], O.BFV)
        g = [oc.apply_galois_coeff(ct[p], elt, ql) for p in range(2)]
        ref = tool.keyswitch_inplace(np.stack([g[0], np.zeros_like(g[0])]), g[1], [glk[i] for i in range(tool.beta)], O.BFV)
        assert np.array_equal(P.to_host(full[b]), ref), b
    del full
    for chunk in (1, 16, 64):
        assert digests(W.relinearize_rotate_batch(ctx, ql, ct3, d_rlk, d_glk, elt, O.BFV, chunk=chunk)) == want, chunk
    for world in (2, 4, 8):
        got = []
        for rank in range(world):
            mine, res = W.relinearize_rotate_sharded(ctx, ql, ct3, d_rlk, d_glk, elt, O.BFV, rank=rank, world=world)
            assert len(mine) == batch // world
            got += digests(res)
        assert got == want, world


def test_config5_128_diagonals_at_the_c3_parameter_set(gpu):
    """BASELINE config 5's building block at its stated size: one 128-diagonal block of the encrypted matrix-vector
    product (127 hoisted rotations + the main diagonal behind ONE mod-up and ONE mod-down) at the CKKS set N = 2^16,
    45 + 15 limbs, against the oracle's hoisting_weighted.  The 127 Galois keys cycle over 3 distinct key buffers and
    the 128 diagonals over 4 distinct plaintexts (the arithmetic does not depend on the data; 127 distinct keys would
    be 23 GB on the host side of the oracle) -- the launches, the pointer tables and the 128-term accumulation are the
    real ones."""
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, n_diag = "c3_ckks16", 45, 128
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(5128)
    elts = [1] + [int(pow(5, k, 2 * n)) for k in range(1, n_diag)]
    key_pool = [_keys(r, primes, n, size_q // size_p) for _ in range(3)]
    qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
    w_pool = [uniform_poly(r, qlp_primes, n) for _ in range(4)]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_key_pool = [P.PhantomRelinKey.from_numpy(k, gpu) for k in key_pool]
    d_w_pool = [P.to_device(w, gpu) for w in w_pool]
    d_keys = [None] + [d_key_pool[k % 3] for k in range(1, n_diag)]
    d_ws = [d_w_pool[k % 4] for k in range(n_diag)]
    out = P.to_host(W.diag_matvec(ctx, ql, P.to_device(ct, gpu), elts, d_keys, d_ws, O.CKKS))
    okeys = [None] + [[key_pool[k % 3][i] for i in range(tool.beta)] for k in range(1, n_diag)]
    want = tool.hoisting_weighted(ct, elts, okeys, [w_pool[k % 4] for k in range(n_diag)], O.CKKS)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("name,scheme,ql,nb,ng,ident", [("hyb12_a2", O.CKKS, 5, 4, 3, True), ("hyb12_a2", O.BGV, 6, 3, 2, True),
                                                        ("hyb13_a3", O.CKKS, 7, 4, 2, False), ("hyb13_a3", O.CKKS, 9, 2, 9, True),
                                                        ("c1_bfv4096", O.CKKS, 2, 3, 2, True)])
def test_config5_bsgs_matches_the_composition_of_reference_steps(name, scheme, ql, nb, ng, ident, gpu):
    """pha_hoisting_weighted_bsgs (baby-step / giant-step, double-hoisted) against the oracle's composition of the restated
    reference steps (Tool.hoisting_weighted_bsgs): identity steps with and without keys around them, a missing term,
    short digits, alpha = 1, bgv."""
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    if scheme == O.BGV:
        ctx.set_plain_modulus(65537)
        tool.set_plain_modulus(65537)
    r = rng_for(950 + nb * 10 + ng)
    first = 0 if ident else 1
    baby = [int(pow(5, k, 2 * n)) for k in range(first, first + nb)]              # ident: element 1 first
    giant = [int(pow(5, nb * k, 2 * n)) for k in range(first, first + ng)]
    dnum = size_q // size_p
    bkeys = [None if e == 1 else _keys(r, primes, n, dnum) for e in baby]
    gkeys = [None if e == 1 else _keys(r, primes, n, dnum) for e in giant]
    qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
    ws = [[uniform_poly(r, qlp_primes, n) for _ in baby] for _ in giant]
    ws[ng - 1][nb - 1] = None                                                     # one missing diagonal
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_bkeys = [None if k is None else P.PhantomRelinKey.from_numpy(k, gpu) for k in bkeys]
    d_gkeys = [None if k is None else P.PhantomRelinKey.from_numpy(k, gpu) for k in gkeys]
    d_ws = [[None if w is None else P.to_device(w, gpu) for w in row] for row in ws]
    d_ct = P.to_device(ct, gpu)
    out = P.to_host(W.diag_matvec_bsgs(ctx, ql, d_ct, baby, d_bkeys, giant, d_gkeys, d_ws, scheme))
    assert np.array_equal(P.to_host(d_ct), ct)
    okb = [None if k is None else [k[i] for i in range(tool.beta)] for k in bkeys]
    okg = [None if k is None else [k[i] for i in range(tool.beta)] for k in gkeys]
    want = tool.hoisting_weighted_bsgs(ct, baby, okb, giant, okg, ws, scheme)
    assert np.array_equal(out, want)
    # one giant step with element 1 is the flat form
    flat = P.to_host(W.diag_matvec_bsgs(ctx, ql, d_ct, baby, d_bkeys, [1], [None], [[w if w is not None else d_ws[0][0] for w in d_ws[0]]], scheme))
    assert np.array_equal(flat, P.to_host(W.diag_matvec(ctx, ql, d_ct, baby, d_bkeys, d_ws[0], scheme)))
    with pytest.raises(ArithmeticError):         # std::logic_error: a rotation without its key (evaluate.cu:1783)
        ctx.hoisting_weighted_bsgs(ql, d_ct.clone(), baby, [None] * nb, giant, d_gkeys, d_ws, scheme)
    # several row blocks against the one ciphertext (shared mod-up, shared pass over the baby keys): every block equals its own call
    ws2 = [[uniform_poly(r, qlp_primes, n) for _ in baby] for _ in giant]
    ws3 = [[uniform_poly(r, qlp_primes, n) for _ in baby] for _ in giant]
    d_blocks = [d_ws, [[P.to_device(w, gpu) for w in row] for row in ws2], [[P.to_device(w, gpu) for w in row] for row in ws3]]
    singles = [out] + [P.to_host(W.diag_matvec_bsgs(ctx, ql, d_ct, baby, d_bkeys, giant, d_gkeys, blk, scheme)) for blk in d_blocks[1:]]
    for per_call in (0, 1, 2, 3):
        multi = P.to_host(W.diag_matvec_bsgs_blocks(ctx, ql, d_ct, baby, d_bkeys, giant, d_gkeys, d_blocks, scheme, per_call=per_call))
        for b in range(3):
            assert np.array_equal(multi[b], singles[b]), (per_call, b)
    assert np.array_equal(P.to_host(d_ct), ct)


def test_config5_bsgs_16_by_8_at_the_c3_parameter_set(gpu):
    """BASELINE config 5 at its stated size in baby-step / giant-step form: 128 diagonals = 16 baby x 8 giant steps at the CKKS set
    N = 2^16, 45 + 15 limbs (15 + 7 Galois keys instead of 127), against the oracle's composition.  Keys cycle over 3 buffers and
    diagonals over 4 plaintexts as in the flat test (the launches, pointer tables and accumulations are the real ones)."""
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, nb, ng = "c3_ckks16", 45, 16, 8
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(5129)
    baby = [int(pow(5, k, 2 * n)) for k in range(nb)]
    giant = [int(pow(5, nb * k, 2 * n)) for k in range(ng)]
    key_pool = [_keys(r, primes, n, size_q // size_p) for _ in range(3)]
    qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
    w_pool = [uniform_poly(r, qlp_primes, n) for _ in range(4)]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_key_pool = [P.PhantomRelinKey.from_numpy(k, gpu) for k in key_pool]
    d_w_pool = [P.to_device(w, gpu) for w in w_pool]
    d_bk = [None] + [d_key_pool[k % 3] for k in range(1, nb)]
    d_gk = [None] + [d_key_pool[(k + 1) % 3] for k in range(1, ng)]
    d_ws = [[d_w_pool[(i * nb + j) % 4] for j in range(nb)] for i in range(ng)]
    out = P.to_host(W.diag_matvec_bsgs(ctx, ql, P.to_device(ct, gpu), baby, d_bk, giant, d_gk, d_ws, O.CKKS))
    okb = [None] + [[key_pool[k % 3][i] for i in range(tool.beta)] for k in range(1, nb)]
    okg = [None] + [[key_pool[(k + 1) % 3][i] for i in range(tool.beta)] for k in range(1, ng)]
    ows = [[w_pool[(i * nb + j) % 4] for j in range(nb)] for i in range(ng)]
    want = tool.hoisting_weighted_bsgs(ct, baby, okb, giant, okg, ows, O.CKKS)
    assert np.array_equal(out, want)


def test_config5_bench_shape_64_by_2_four_row_blocks_per_call(gpu):
    """BASELINE config 5 at the shape bench.py's `matvec_c5` leg runs: 128 diagonals = 64 baby x 2 giant steps at
    the CKKS set N = 2^16, 45 + 15 limbs, FOUR row blocks per call of pha_hoisting_weighted_bsgs_blocks (8 (block, giant step)
    accumulators: the full width of the fused baby-step kernel, one pass over the 63 baby keys for the four blocks) -- every block
    against the oracle's composition (Tool.hoisting_weighted_bsgs, src/evaluate.cu:1670-1866 steps).  Keys cycle over 3 buffers and
    the diagonals over a pool of 5 plaintexts taken at a different offset in every block, as bench.py rotates its pool (the launches,
    pointer tables and accumulations are the real ones; the oracle's cost stays at ~25 s per block)."""
    import phantom_fhe_amd as P
    from phantom_fhe_amd import workloads as W
    name, ql, nb, ng, nblk = "c3_ckks16", 45, 64, 2, 4
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc, ctx = oracle_ctx(name), _ctx(name, gpu)
    tool = O.Tool(oc, ql)
    r = rng_for(5130)
    baby = [int(pow(5, k, 2 * n)) for k in range(nb)]
    giant = [int(pow(5, nb * k, 2 * n)) for k in range(ng)]
    key_pool = [_keys(r, primes, n, size_q // size_p) for _ in range(3)]
    qlp_primes = [primes[i] for i in list(range(ql)) + [size_q + j for j in range(size_p)]]
    w_pool = [uniform_poly(r, qlp_primes, n) for _ in range(5)]
    ct = np.stack([uniform_poly(r, primes[:ql], n) for _ in range(2)])
    d_key_pool = [P.PhantomRelinKey.from_numpy(k, gpu) for k in key_pool]
    d_w_pool = [P.to_device(w, gpu) for w in w_pool]
    d_bk = [None] + [d_key_pool[k % 3] for k in range(1, nb)]
    d_gk = [None] + [d_key_pool[(k + 1) % 3] for k in range(1, ng)]
    pick = lambda b, i, j: (i * nb + j + b) % 5          # block b takes the pool rotated by b
    d_blocks = [[[d_w_pool[pick(b, i, j)] for j in range(nb)] for i in range(ng)] for b in range(nblk)]
    d_ct = P.to_device(ct, gpu)
    out = P.to_host(W.diag_matvec_bsgs_blocks(ctx, ql, d_ct, baby, d_bk, giant, d_gk, d_blocks, O.CKKS, per_call=4))
    assert np.array_equal(P.to_host(d_ct), ct)                                  # the input ciphertext is only read
    okb = [None] + [[key_pool[k % 3][i] for i in range(tool.beta)] for k in range(1, nb)]
    okg = [None] + [[key_pool[(k + 1) % 3][i] for i in range(tool.beta)] for k in range(1, ng)]
    for b in range(nblk):
        ows = [[w_pool[pick(b, i, j)] for j in range(nb)] for i in range(ng)]
        want = tool.hoisting_weighted_bsgs(ct, baby, okb, giant, okg, ows, O.CKKS)
        assert np.array_equal(out[b], want), b
    # r04: the bench now sends EIGHT row blocks through one call (16 (block, giant step) accumulators: the FP64 limbs keep all 16 in
    # registers, the integer limbs walk the baby steps twice with 8 each).  Blocks 0..3 of such a call must equal the oracle-checked
    # words above, blocks 4..7 their own four-block call.
    d_blocks8 = d_blocks + [[[d_w_pool[pick(b, i, j)] for j in range(nb)] for i in range(ng)] for b in range(nblk, 2 * nblk)]
    out8 = P.to_host(W.diag_matvec_bsgs_blocks(ctx, ql, d_ct, baby, d_bk, giant, d_gk, d_blocks8, O.CKKS, per_call=8))
    assert np.array_equal(out8[:nblk], out)
    tail4 = P.to_host(W.diag_matvec_bsgs_blocks(ctx, ql, d_ct, baby, d_bk, giant, d_gk, d_blocks8[nblk:], O.CKKS, per_call=4))
    assert np.array_equal(out8[nblk:], tail4)
