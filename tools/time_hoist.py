"""Timing of hoisted rotations vs single rotations at the C3 parameter set (development helper, GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
for _kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):
    P.set_tuning(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
from util import primes_of

log_n, primes, size_p = primes_of("c3_ckks16")
n = 1 << log_n
size_q = len(primes) - size_p
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, list(primes), size_p, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1)
def rnd(shape_limbs):
    out = torch.empty((len(shape_limbs), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(shape_limbs):
        out[i] = torch.randint(0, int(q), (n,), generator=gen, device=dev, dtype=torch.int64)
    return out
dnum = size_q // size_p
def key():
    return P.PhantomRelinKey([torch.stack([rnd(primes), rnd(primes)]) for _ in range(dnum)])
for n_elts in (1, 4, 8, 16):
    elts = [pow(5, i + 1, 2 * n) for i in range(n_elts)]
    keys = [key() for _ in elts]
    ct = torch.stack([rnd(primes[:size_q]), rnd(primes[:size_q])])
    for _ in range(3):
        ctx.hoisting(size_q, ct, elts, keys, P.scheme_type.ckks)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.hoisting(size_q, ct, elts, keys, P.scheme_type.ckks)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"hoisting of {n_elts:2d} rotations: {ms:.3f} ms per call, {ms / n_elts:.3f} ms per rotation")
    del keys
