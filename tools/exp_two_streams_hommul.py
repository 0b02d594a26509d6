"""r05: does running TWO half-batches of the batched HomMul + relinearize + rescale on two streams beat one batch on one stream?  (The op alternates
issue-bound kernels -- conversions, transforms -- with memory-bound ones -- tensor product, inner product; two op sets in flight could overlap them.)"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bt1, bt2 = rnd(B, 2, ql, n), rnd(B, 2, ql, n)
b01, b2 = torch.zeros_like(bt1), torch.zeros((B, ql, n), dtype=torch.int64, device=dev)
bout = torch.zeros((B, 2, ql - 1, n), dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

def one(lo, hi):
    ctx.tensor_prod_2x2_batched(bt1[lo:hi], bt2[lo:hi], b01[lo:hi], b2[lo:hi], ql, hi - lo)
    ctx.keyswitch_rescale_batched(ql, b01[lo:hi], b2[lo:hi], hi - lo, rlk.public_keys_ptr, bout[lo:hi])

def single():
    one(0, B)

def two(stagger):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        one(0, B // 2)
    with torch.cuda.stream(s2):
        one(B // 2, B)
    cur.wait_stream(s1); cur.wait_stream(s2)

def timed(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / B * 1e3

single(); chk1 = int(bout.sum().item())
bout.zero_(); two(0); chk2 = int(bout.sum().item())
for rep in range(3):
    print(f"B = {B}: one stream {timed(single):7.1f} us/op | two streams, half each {timed(lambda: two(0)):7.1f} us/op | same words {chk1 == chk2}", flush=True)
