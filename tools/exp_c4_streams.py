"""Config 4 (BFV relinearize + rotate, N = 2^15, 30 + 15 limbs, 64 ciphertexts): the one-call form on one stream against the same
work split over two / four HIP streams (each half through its own call; development experiment)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from phantom_fhe_amd import workloads as W
from util import primes_of

log_n, primes, size_p = primes_of("c4_bfv15")
n = 1 << log_n
size_q = len(primes) - size_p
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, list(primes), size_p, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(4)
def rnd(*shape):
    return torch.randint(0, 1 << 49, shape, generator=gen, device=dev, dtype=torch.int64)
dnum = size_q // size_p
rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(dnum)])
glk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(dnum)])
BFV = P.scheme_type.bfv
ct3 = rnd(64, 3, size_q, n)
ref = W.relinearize_rotate_batch(ctx, size_q, ct3, rlk, glk, 3, BFV)
torch.cuda.synchronize()
def whole(chunk):
    def run():
        return W.relinearize_rotate_batch(ctx, size_q, ct3, rlk, glk, 3, BFV, chunk=chunk)
    out = run(); torch.cuda.synchronize()
    assert torch.equal(out, ref)
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"one call, chunk argument {chunk} (0 = two internal streams x half a set): {ms:7.2f} ms, {64 / ms * 1e3:7.0f} ciphertexts/s")
for _ in range(3):
    whole(0); whole(8)
for nstreams, chunk in ((1, 8), (2, 4), (1, 8), (2, 4)):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    parts = [ct3[i * 64 // nstreams:(i + 1) * 64 // nstreams].contiguous() for i in range(nstreams)]
    outs = [None] * nstreams
    def run():
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = W.relinearize_rotate_batch(ctx, size_q, parts[i], rlk, glk, 3, BFV, chunk=chunk)
    run(); torch.cuda.synchronize()
    assert torch.equal(torch.cat(outs), ref)
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{nstreams} stream(s), sub-batches of {chunk}: {ms:7.2f} ms per 64 ciphertexts, {64 / ms * 1e3:7.0f} ciphertexts/s")
