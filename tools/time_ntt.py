"""Quick hipEvent timing of the forward NTT at the BASELINE configs (development helper)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from util import primes_of, rng_for, uniform_poly

for name, limbs in [("c2_ntt14", 8), ("c4_bfv15", 30), ("c3_ckks16", 45), ("c3_ckks16", 60), ("c3_ckks16", 1)]:
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    ctx = P.PhantomContext(log_n, list(primes), size_p, device=0)
    x = P.to_device(uniform_poly(rng_for(1), primes[:limbs], n), "cuda:0")
    ctx.time_forward_ntt(x, limbs, 20)
    best = min(ctx.time_forward_ntt(x, limbs, 200) for _ in range(5))
    by = 16.0 * n * limbs
    print(f"{name} N=2^{log_n} limbs={limbs}: {best*1e3:.2f} us/launch-pair  {limbs/(best*1e-3):.0f} limb-NTT/s  "
          f"{by/(best*1e-3)/1e12:.3f} TB/s algorithmic ({by/(best*1e-3)/8e12*100:.1f}% of 8 TB/s)")
