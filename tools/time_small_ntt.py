"""Forward / inverse NTT of small launches (1 .. 60 limbs of one polynomial, N = 2^16 or PHA_OPS_LOGN) in place, mean GPU time of
200 launches from one event pair: the latency regime of the transform (development helper; A/B of builds via PHA_LIB_OVERRIDE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

log_n = int(os.environ.get("PHA_OPS_LOGN", "16"))
n = 1 << log_n
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 59)]
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, primes, 0, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
buf = torch.randint(0, 1 << 49, (16, 60, n), dtype=torch.int64, device=dev, generator=gen)   # 16 buffers: rotate (HBM / MALL mix)
def t(fn, reps=200):
    for _ in range(20): fn(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print(f"N = 2^{log_n}; us per launch pair (forward / inverse), one polynomial of L limbs, rotating over 16 buffers")
for L in (1, 2, 8, 16, 24, 30, 32, 40, 45, 52, 60):
    f = t(lambda i: ctx.nwt_2d_radix8_forward_inplace(buf[i % 16], L, 0))
    v = t(lambda i: ctx.nwt_2d_radix8_backward_inplace(buf[i % 16], L, 0))
    print(f"L = {L:3d}: {f:7.2f} {v:7.2f}")
