"""The bench's headline step alone (forward NTT of 16 x 45 limbs at N = 2^16, 40 steps) -- the workload of per-kernel traces of
experiment builds (PHA_LIB_OVERRIDE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import phantom_fhe_amd as P
n = 1 << 16
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44 + [60] * 15)]
ctx = P.PhantomContext(16, primes, 15, device=torch.device("cuda:0"))
x = torch.randint(0, 1 << 49, (16, 45, n), dtype=torch.int64, device="cuda:0")
for _ in range(40):
    ctx.nwt_2d_radix8_forward_inplace_batched(x, 45, 0, 16, 45 * n)
torch.cuda.synchronize()
