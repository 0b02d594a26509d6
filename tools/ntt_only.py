"""Tiny NTT-only workload for PMC passes (GPU box): a few 45-limb and 4 x 45-limb forward transforms at N = 2^16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from util import primes_of
log_n, primes, size_p = primes_of("c3_ckks16")
n = 1 << log_n
ctx = P.PhantomContext(log_n, list(primes), size_p, device=0)
x = torch.randint(0, 1 << 39, (4, 45, n), dtype=torch.int64, device="cuda:0")
for _ in range(4):
    ctx.nwt_2d_radix8_forward_inplace(x[0], 45, 0)
    ctx.nwt_2d_radix8_forward_inplace_batched(x, 45, 0, 4, 45 * n)
torch.cuda.synchronize()
