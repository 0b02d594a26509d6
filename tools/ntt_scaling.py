"""Launch forward NTTs of different sizes (run under rocprofv3 --kernel-trace to get GPU-side durations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phantom_fhe_amd as P
from util import primes_of, rng_for, uniform_poly
log_n, primes, size_p = primes_of("c3_ckks16")
n = 1 << log_n
ctx = P.PhantomContext(log_n, list(primes), size_p, device=0)
x = P.to_device(np.stack([uniform_poly(rng_for(1), primes[:45], n) for _ in range(4)]), "cuda:0")
for variant in (1, 17):
    P.set_tuning(0, variant)
    for limbs, batch in [(1, 1), (4, 1), (16, 1), (45, 1), (45, 2), (45, 4)]:
        for _ in range(12):
            ctx.nwt_2d_radix8_forward_inplace_batched(x, limbs, 0, batch, 45 * n)
        torch.cuda.synchronize()
