"""Phase stamps of the MFMA GEMM (library built with -DPHA_GEMM_X=5: s_memtime at start / after the prologue / after every pair of
blocks / after the reduction, written over C).  Development helper."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import numpy as np
import torch
import phantom_fhe_amd as P
NB = int(os.environ.get("GEMM_BATCH", "30"))
primes = [int(p) for p in P.coeff_modulus_create(4096, [50] * NB)]
ctx = P.PhantomContext(12, primes, 0, device=0)
m = n = k = 256
A = torch.randint(0, 1 << 40, (NB, m, k), dtype=torch.int64, device="cuda:0")
B = torch.randint(0, 1 << 40, (NB, k, n), dtype=torch.int64, device="cuda:0")
C = torch.zeros((NB, m, n), dtype=torch.int64, device="cuda:0")
for _ in range(3):
    ctx.batched_modular_gemm(C, A, B, m, n, k, NB)
torch.cuda.synchronize()
st = C.flatten()[: NB * 16 * 8].reshape(-1, 8).cpu().numpy().astype("int64")
t0 = st[:, 0].min()
d = np.diff(st[:, :7], axis=1)
print("tiles", len(st), " ticks per phase [prologue, pair0, pair1, pair2, pair3, reduce]")
print("mean  ", d.mean(axis=0).round(0))
print("median", np.median(d, axis=0))
print("start spread (0/50/90/100 %)", np.percentile(st[:, 0] - t0, [0, 50, 90, 100]))
print("end   spread (0/50/90/100 %)", np.percentile(st[:, 6] - t0, [0, 50, 90, 100]))
for i in (0, 1, 500, 1000, 1500, len(st) - 1):
    if i < len(st):
        print(i, (st[i, :7] - t0).tolist())
