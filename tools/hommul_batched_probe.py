"""A few batched HomMul + relinearize + rescale ops (B ciphertext pairs per launch set, C3 set) -- the workload of a per-kernel trace of
the batched entry points (pha_tensor_prod_2x2_batched + pha_keyswitch_rescale_batched).  usage: hommul_batched_probe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import phantom_fhe_amd as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << 16
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44 + [60] * 15)]
ql = 45
dev = torch.device("cuda:0")
ctx = P.PhantomContext(16, primes, 15, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = lambda *s: torch.randint(0, 1 << 49, s, dtype=torch.int64, device=dev, generator=g)
rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(3)])
bt1, bt2 = rnd(B, 2, ql, n), rnd(B, 2, ql, n)
b01, b2, bout = torch.zeros_like(bt1), torch.zeros((B, ql, n), dtype=torch.int64, device=dev), torch.zeros((B, 2, ql - 1, n), dtype=torch.int64, device=dev)
for _ in range(5):
    ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, ql, B)
    ctx.keyswitch_rescale_batched(ql, b01, b2, B, rlk.public_keys_ptr, bout)
torch.cuda.synchronize()
