"""A few single HomMul + relinearize + rescale ops (C3 set) -- the workload of a kernel-trace timeline (tools/hommul_single_timeline.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import phantom_fhe_amd as P
n = 1 << 16
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44 + [60] * 15)]
ql = 45
dev = torch.device("cuda:0")
ctx = P.PhantomContext(16, primes, 15, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = lambda *s: torch.randint(0, 1 << 49, s, dtype=torch.int64, device=dev, generator=g)
rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(3)])
ct1, ct2 = rnd(2, ql, n), rnd(2, ql, n)
buf = torch.zeros((3, ql, n), dtype=torch.int64, device=dev)
out = torch.zeros((2, ql - 1, n), dtype=torch.int64, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ctx.tensor_prod_2x2_rns_poly(ct1, ct2, buf, ql)
    ctx.keyswitch_rescale(ql, buf, buf[2], rlk.public_keys_ptr, out)
torch.cuda.synchronize()
