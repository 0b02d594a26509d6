"""Batched CKKS key switch WITHOUT rescale at the C3 set (pha_keyswitch_inplace_batched): us per key switch at B = 8 / 32 with a checksum of the
outputs (A/B of builds through PHA_LIB_OVERRIDE)."""
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
out = []
for B in [8, 32]:
    ct, c2 = rnd(B, 2, ql, n), rnd(B, ql, n)
    ct0 = ct.clone()
    def step():
        ctx.keyswitch_inplace_batched(ql, ct, c2, B, rlk.public_keys_ptr, P.scheme_type.ckks)
    best = 1e9
    for rep in range(3):
        for _ in range(2): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8 / B)
    ct.copy_(ct0); step()
    out.append(f"B={B}: {best * 1e3:6.1f} us/keyswitch chk {int(ct.sum().item()) & 0xffffffff:x}")
print(" | ".join(out))
