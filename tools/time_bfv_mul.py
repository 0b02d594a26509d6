"""Timing of BFV multiply (BEHZ and HPS) + relinearize at the C4 parameter set (N = 2^15, 30 + 15 limbs), GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from util import primes_of
log_n, primes, size_p = primes_of("c4_bfv15")
n = 1 << log_n
size_q = len(primes) - size_p
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, list(primes), size_p, device=dev)
ctx.set_plain_modulus(1032193)
gen = torch.Generator(device=dev); gen.manual_seed(1)
def rnd(limbs):
    out = torch.empty((len(limbs), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(limbs):
        out[i] = torch.randint(0, int(q), (n,), generator=gen, device=dev, dtype=torch.int64)
    return out
ct1 = torch.stack([rnd(primes[:size_q]) for _ in range(2)])
ct2 = torch.stack([rnd(primes[:size_q]) for _ in range(2)])
dst = torch.zeros((3, size_q, n), dtype=torch.int64, device=dev)
rlk = P.PhantomRelinKey([torch.stack([rnd(primes), rnd(primes)]) for _ in range(size_q // size_p)])
for name, fn in (("BEHZ", ctx.bfv_multiply_behz), ("HPS", ctx.bfv_multiply_hps)):
    for _ in range(3):
        fn(ct1, ct2, dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn(ct1, ct2, dst)
    torch.cuda.synchronize()
    mul_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(ct1, ct2, dst)
        ctx.keyswitch_inplace(size_q, dst[:2], dst[2], rlk.public_keys_ptr, P.scheme_type.bfv)
    torch.cuda.synchronize()
    both_ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"BFV {name} multiply at N=2^15, 30 limbs: {mul_ms:.3f} ms; multiply + relinearize: {both_ms:.3f} ms")
