"""Quick hipEvent timing of the forward NTT at the BASELINE configs (development helper)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from util import primes_of, rng_for, uniform_poly

CASES = [("c2_ntt14", 8), ("c4_bfv15", 30), ("c3_ckks16", 1), ("c3_ckks16", 45), ("c3_ckks16", 60)]
ctxs = {}
for name, limbs in CASES:
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    if name not in ctxs:
        ctxs[name] = P.PhantomContext(log_n, list(primes), size_p, device=0)
    ctx = ctxs[name]
    reps = (limbs + len(primes) - 1) // len(primes)
    # limbs > #primes: several polynomials back to back would need a limb map; time a 60-limb launch x reps instead
    lim = min(limbs, len(primes))
    x = P.to_device(uniform_poly(rng_for(1), primes[:lim], n), "cuda:0")
    for variant in (33, 97):
        P.set_tuning(0, variant)
        ctx.time_forward_ntt(x, lim, 20)
        best = min(ctx.time_forward_ntt(x, lim, 200) for _ in range(5))
        by = 16.0 * n * lim
        print(f"v{variant} {name} N=2^{log_n} limbs={lim}: {best*1e3:.2f} us/launch-pair  {lim/(best*1e-3):.0f} limb-NTT/s  "
              f"{by/(best*1e-3)/1e12:.3f} TB/s algorithmic ({by/(best*1e-3)/8e12*100:.1f}% of 8 TB/s)")
P.set_tuning(0, 0)
