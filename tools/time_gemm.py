"""Timing of the batched modular GEMM at the reference benchmark's shape (256^3 x 30 moduli of 50 bits), GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P
for bits in (50, 60):
    primes = [int(p) for p in P.coeff_modulus_create(4096, [bits] * 30)]
    ctx = P.PhantomContext(12, primes, 0, device=0)
    m = n = k = 256
    A = torch.randint(0, 1 << 40, (30, m, k), dtype=torch.int64, device="cuda:0")
    B = torch.randint(0, 1 << 40, (30, k, n), dtype=torch.int64, device="cuda:0")
    C = torch.zeros((30, m, n), dtype=torch.int64, device="cuda:0")
    for _ in range(5):
        ctx.batched_modular_gemm(C, A, B, m, n, k, 30)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.batched_modular_gemm(C, A, B, m, n, k, 30)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    macs = 30 * m * n * k
    print(f"{bits}-bit moduli: {us:.1f} us per 30 x 256^3 batch, {macs / us / 1e6:.2f} T modular MAC/s, "
          f"{4 * macs / us / 1e6:.1f} T v_mad_u64_u32 lane-ops/s")
