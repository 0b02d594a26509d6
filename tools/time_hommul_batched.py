"""Batched HomMul + relinearize + rescale at the C3 set (pha_tensor_prod_2x2_batched + pha_keyswitch_rescale_batched): ms per op for
B in argv (default 2 4 8 16), event pairs on the launch stream; a checksum of the outputs so that builds / switches can be compared."""
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
out_line = []
for B in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]:
    bt1, bt2 = rnd(B, 2, ql, n), rnd(B, 2, ql, n)
    b01, b2 = torch.zeros_like(bt1), torch.zeros((B, ql, n), dtype=torch.int64, device=dev)
    bout = torch.zeros((B, 2, ql - 1, n), dtype=torch.int64, device=dev)
    def step():
        ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, ql, B)
        ctx.keyswitch_rescale_batched(ql, b01, b2, B, rlk.public_keys_ptr, bout)
    best = 1e9
    for rep in range(3):
        for _ in range(3): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 / B)
    out_line.append(f"B={B}: {best * 1e3:6.1f} us/op chk {int(bout.sum().item()) & 0xffffffff:x}")
print(" | ".join(out_line))
