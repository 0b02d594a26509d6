"""Timing of BASELINE config 4 on one GPU: BFV relinearize + Galois rotate over a batch of 64 ciphertexts at N = 2^15,
30 + 15 limbs (workloads.relinearize_rotate_batch), whole batch and in sub-batches (development helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
for _kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):
    P.set_tuning(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
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
for sub in (0, 8, 0, 8, 4, 16):     # 0 = the library's own choice (two internal streams, half a set each)     # ciphertexts per set of launches inside the one C call (pha_relinearize_rotate_batched's chunk)
    def run():
        W.relinearize_rotate_batch(ctx, size_q, ct3, rlk, glk, 3, BFV, chunk=sub)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"64 ciphertexts in sub-batches of {sub:2d}: {ms:8.2f} ms  ({ms / 64 * 1e3:7.1f} us per relinearize + rotate, {64 / ms * 1e3:8.0f} ciphertexts/s)")
