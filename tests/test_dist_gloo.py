"""world_size-2 gloo test of the multi-GPU host logic (runs on CPU): contiguous ciphertext sharding
with no data-path collective, one-time key broadcast, max-over-ranks timing, checksum gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "phantom-fhe_amd"))
    from phantom_fhe_amd import dist as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # keys exist only on rank 0 before the broadcast
        g = torch.Generator().manual_seed(1234)
        keys = [torch.randint(0, 1 << 50, (2, 3, 64), dtype=torch.int64, generator=g) if rank == 0
                else torch.zeros((2, 3, 64), dtype=torch.int64) for _ in range(2)]
        assert pd.broadcast_keys(keys, src=0) == 2          # separately allocated tensors: one call each
        key_sum = int(sum(int(k.sum()) for k in keys))
        # a key set cut out of ONE allocation (what bench.py does) travels as one flat buffer, split only at the chunk size
        slab = torch.randint(0, 1 << 50, (5, 3, 2, 3, 64), dtype=torch.int64, generator=g) if rank == 0 \
            else torch.zeros((5, 3, 2, 3, 64), dtype=torch.int64)
        views = [slab[k][d] for k in range(5) for d in range(3)]
        assert pd.broadcast_keys(views, src=0) == 1
        key_sum += int(slab.sum())
        slab2 = slab * 3 if rank == 0 else torch.zeros_like(slab)
        pd.BROADCAST_CHUNK_BYTES = slab2[0].numel() * 8 * 2        # two keys per call: 5 keys -> 3 calls
        assert pd.broadcast_keys([slab2[k][d] for k in range(5) for d in range(3)], src=0) == 3
        key_sum += int(slab2.sum())
        # direct=True without a reachable RCCL communicator (here: gloo, host tensors) falls back LOUDLY, with the reason
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            assert pd.broadcast_keys(keys, src=0, ctx=object(), direct=True) == 2
        assert pd.LAST_BROADCAST_PATH == "dist.broadcast"
        assert pd.LAST_BROADCAST_FALLBACK and "HIP device" in pd.LAST_BROADCAST_FALLBACK
        assert any("falling back to dist.broadcast" in str(w.message) and issubclass(w.category, RuntimeWarning) for w in caught)
        assert pd.broadcast_keys(keys, src=0) == 2 and pd.LAST_BROADCAST_FALLBACK is None
        comm, why = pd._raw_nccl_comm("cpu")
        assert comm is None and "gloo" in why
        # ranks that lay the same key set out differently (one slab here, separate tensors there) would issue different
        # collective sequences: the call refuses on every rank instead of hanging
        bad = [slab[0][0], slab[0][1]] if rank == 0 else [slab[0][0].clone(), slab[0][1].clone()]
        try:
            pd.broadcast_keys(bad, src=0)
            refused = False
        except RuntimeError as e:
            refused = "different collective sequences" in str(e)
        assert refused
        mine = list(pd.shard_range(batch, rank, world))
        # "process" the shard: a per-ciphertext function of the index and the key only (no cross-rank data)
        local = sum((i * 2654435761 + key_sum) % (1 << 40) for i in mine)
        sums = pd.gather_checksums(local)
        t = pd.max_over_ranks(0.5 + rank)
        q.put((rank, mine, key_sum, sums, t))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_key_broadcast():
    world, batch = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, k0, sums0, t0), (r1, s1, k1, sums1, t1) = res
    assert s0 + s1 == list(range(batch)) and abs(len(s0) - len(s1)) <= 1     # disjoint contiguous cover
    assert k0 == k1                                                           # broadcast reached rank 1
    assert sums0 == sums1 and len(sums0) == 2
    single = sum((i * 2654435761 + k0) % (1 << 40) for i in range(batch))
    assert sum(sums0) == single                                               # 2-rank result == 1-rank result
    assert t0 == t1 == 1.5                                                    # max over ranks


def test_shard_range_properties():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "phantom-fhe_amd"))
    from phantom_fhe_amd.dist import shard_range
    for batch in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [list(shard_range(batch, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(batch))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


class _StubKey:
    public_keys_ptr = None


class _StubCtx:
    """Stands in for PhantomContext on the CPU: deterministic tensor functions with the same call shapes, so that
    only the host logic of phantom_fhe_amd.workloads (sharding, copies, aliasing) is under test here."""

    def keyswitch_inplace_batched(self, ql, ct, c2, batch, keys, scheme):
        assert ct.shape[0] == batch == c2.shape[0]
        ct += c2[:, None] * 3

    def apply_galois(self, src, dst, elt, ql, mod_start=0):
        dst.copy_(src.flip(-1) + elt)

    apply_galois_ntt = apply_galois

    def apply_galois_batched(self, src, dst, elt, ql, polys, ntt_form):
        dst.copy_(src.flip(-1) + elt)

    def apply_galois_for_keyswitch(self, src, dst_ct, dst_c2, elt, ql, batch, ntt_form):
        g = src.flip(-1) + elt
        dst_ct[:, 0] = g[:, 0]
        dst_ct[:, 1] = 0
        dst_c2.copy_(g[:, 1])

    def relinearize_rotate_batched(self, ql, ct3, batch, rlk, glk, elt, scheme, out, chunk=0):
        assert ct3.shape[0] == batch == out.shape[0]
        ct = ct3[:, :2] + ct3[:, 2][:, None] * 3          # the stub key switch
        g = ct.flip(-1) + elt                             # the stub automorphism
        out[:, 0] = g[:, 0]
        out[:, 1] = 0
        out += g[:, 1][:, None] * 3

    def hoisting_weighted(self, ql, ct, elts, keys, weights, scheme):
        acc = sum(w[:ql] * int(e) for w, e in zip(weights, elts))
        ct *= acc[None]


def _workload_worker(rank, world, port, q, batch=5, nblocks=5):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "phantom-fhe_amd"))
    from phantom_fhe_amd import workloads as W
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(99)
        ql, n = 3, 16
        ct3 = torch.randint(0, 1 << 40, (batch, 3, ql, n), dtype=torch.int64, generator=g)
        keep = ct3.clone()
        ctx = _StubCtx()
        mine, res = W.relinearize_rotate_sharded(ctx, ql, ct3, _StubKey(), _StubKey(), 3, 1)   # rank / world from the group
        assert torch.equal(ct3, keep)                                                         # inputs are not consumed
        ct = torch.randint(0, 1 << 20, (2, ql, n), dtype=torch.int64, generator=g)
        blocks = [[torch.randint(0, 1 << 20, (ql + 2, n), dtype=torch.int64, generator=g) for _ in range(3)] for _ in range(nblocks)]
        mine5, outs = W.matvec_row_blocks_sharded(ctx, ql, ct, [1, 5, 25], [None, _StubKey(), _StubKey()], blocks, 2)
        gathered = [None] * world
        dist.all_gather_object(gathered, (list(mine), res, list(mine5), outs))
        if rank == 0:
            full = W.relinearize_rotate_batch(ctx, ql, ct3, _StubKey(), _StubKey(), 3, 1)
            _, full5 = W.matvec_row_blocks_sharded(ctx, ql, ct, [1, 5, 25], [None, _StubKey(), _StubKey()], blocks, 2, rank=0, world=1)
            idx = sum((g_[0] for g_ in gathered), [])
            ok4 = idx == list(range(batch)) and torch.equal(torch.cat([g_[1] for g_ in gathered]), full)
            idx5 = sum((g_[2] for g_ in gathered), [])
            outs5 = sum((g_[3] for g_ in gathered), [])
            ok5 = idx5 == list(range(nblocks)) and all(torch.equal(a, b) for a, b in zip(outs5, full5))
            # the checksum of checksums bench.py prints (sum mod 2^64 of all output words) does not depend on the world size
            from phantom_fhe_amd import dist as pd
            q.put((ok4, ok5))
        from phantom_fhe_amd import dist as pd
        local = int(res.sum().item()) if len(mine) else 0
        sums = pd.gather_checksums(local)
        whole = int(W.relinearize_rotate_batch(ctx, ql, ct3, _StubKey(), _StubKey(), 3, 1).sum().item())
        assert len(sums) == world and sum(sums) == whole, (sums, whole)
        # a key slab generated on rank 0 reaches every rank in ONE collective call
        slab = torch.arange(4 * 2 * 5 * n, dtype=torch.int64).reshape(4, 2, 5, n) * (1 if rank == 0 else 0)
        calls = pd.broadcast_keys([slab[i] for i in range(4)], src=0)
        assert calls == 1 and torch.equal(slab, torch.arange(4 * 2 * 5 * n, dtype=torch.int64).reshape(4, 2, 5, n))
    finally:
        dist.destroy_process_group()


def test_two_rank_config4_and_config5_sharding_reproduces_one_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_workload_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok4, ok5 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok4 and ok5


def _gpu_workload_worker(rank, world, port, q):
    """The real PhantomContext on cuda:0 in every rank (the GPU box has one device; gloo carries the key broadcast and the
    gather): config-4 composition on this rank's shard with keys that exist only on rank 0 before the broadcast."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "phantom-fhe_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    import phantom_fhe_amd as P
    from phantom_fhe_amd import dist as pd
    from phantom_fhe_amd import workloads as W
    from oracle import oracle as O
    from util import oracle_ctx, primes_of, rng_for, uniform_poly
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        name, ql, batch, elt = "hyb12_a2", 6, 5, 3
        log_n, primes, size_p = primes_of(name)
        n = 1 << log_n
        size_q = len(primes) - size_p
        dev = torch.device("cuda:0")
        ctx = P.PhantomContext(log_n, list(primes), size_p, device=dev)
        r = rng_for(4242)                                   # same stream in every rank: inputs are common knowledge ...
        ct3 = np.stack([np.stack([uniform_poly(r, primes[:ql], n) for _ in range(3)]) for _ in range(batch)])
        keys = [np.stack([uniform_poly(r, primes, n), uniform_poly(r, primes, n)]) for _ in range(2 * (size_q // size_p))]
        host = [torch.from_numpy(k.view(np.int64)) if rank == 0 else torch.zeros((2, len(primes), n), dtype=torch.int64) for k in keys]
        pd.broadcast_keys(host, src=0)                      # ... the keys are not: ranks > 0 only have them after this
        d_keys = [h.to(dev) for h in host]
        half = len(d_keys) // 2
        rlk, glk = P.PhantomRelinKey(d_keys[:half]), P.PhantomRelinKey(d_keys[half:])
        mine, res = W.relinearize_rotate_sharded(ctx, ql, P.to_device(ct3, dev), rlk, glk, elt, O.BFV)
        gathered = [None] * world
        dist.all_gather_object(gathered, (list(mine), P.to_host(res)))
        if rank == 0:
            idx = sum((g[0] for g in gathered), [])
            got = np.concatenate([g[1] for g in gathered])
            oc = oracle_ctx(name)
            tool = O.Tool(oc, ql)
            ok = idx == list(range(batch))
            for b in range(batch):
                ct = tool.keyswitch_inplace(ct3[b, :2], ct3[b, 2], [keys[i] for i in range(tool.beta)], O.BFV)
                g = [oc.apply_galois_coeff(ct[p], elt, ql) for p in range(2)]
                want = tool.keyswitch_inplace(np.stack([g[0], np.zeros_like(g[0])]), g[1], [keys[half + i] for i in range(tool.beta)], O.BFV)
                ok = ok and np.array_equal(got[b], want)
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


def test_eight_rank_config4_and_config5_sharding_reproduces_one_rank():
    """The world size of the node the scaling bench runs on: 8 gloo ranks, an uneven batch (19 ciphertexts: shards of
    3 and 2) and fewer row blocks than ranks (5: three ranks own nothing) -- index cover, results, checksum of checksums and the one-call
    key broadcast all as in the two-rank test."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_workload_worker, args=(r, world, port, q, 19, 5)) for r in range(world)]
    for p in procs:
        p.start()
    ok4, ok5 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok4 and ok5


@pytest.mark.gpu
def test_two_ranks_with_the_real_context_reproduce_the_oracle(gpu):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_workload_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok
