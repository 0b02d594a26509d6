"""Timing of the batched modular GEMM at the reference benchmark's shape (256^3 x 30 moduli of 50 bits), GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P
NB = int(os.environ.get("GEMM_BATCH", "30"))       # moduli (the reference benchmark: 30)
for bits in (50, 60):
    primes = [int(p) for p in P.coeff_modulus_create(4096, [bits] * NB)]
    ctx = P.PhantomContext(12, primes, 0, device=0)
    m = n = k = 256
    A = torch.randint(0, 1 << 40, (NB, m, k), dtype=torch.int64, device="cuda:0")
    B = torch.randint(0, 1 << 40, (NB, k, n), dtype=torch.int64, device="cuda:0")
    C = torch.zeros((NB, m, n), dtype=torch.int64, device="cuda:0")
    for _ in range(5):
        ctx.batched_modular_gemm(C, A, B, m, n, k, NB)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.batched_modular_gemm(C, A, B, m, n, k, NB)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    macs = NB * m * n * k
    digits = 7 if bits <= 50 else 8                    # signed bytes per operand: digits^2 i8 multiply-adds per modular one
    i8 = macs * digits * digits
    chk = int(C.sum().item()) & 0xffffffff
    print(f"{bits}-bit moduli: chk {chk:08x} {us:.1f} us per {NB} x 256^3 batch, {macs / us / 1e6:.2f} T modular MAC/s, "
          f"{i8 / us / 1e9:.2f} P i8 MAC/s = {i8 / us / 1e9 / 2.2 * 100:.0f} % of the measured i8 MFMA rate (4.4 POP/s)")
