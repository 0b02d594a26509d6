"""Timing of one 128-diagonal block of BASELINE config 5 at the C3 set (N = 2^16, 45 + 15 limbs) in baby-step / giant-step form
(pha_hoisting_weighted_bsgs) for several (baby, giant) splits, next to the flat hoisted form (development helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
from phantom_fhe_amd import workloads as W
from util import primes_of

log_n, primes, size_p = primes_of("c3_ckks16")
n = 1 << log_n
size_q = len(primes) - size_p
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, list(primes), size_p, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1)
def rnd(limbs):
    return torch.randint(0, 1 << 49, (limbs, n), generator=gen, device=dev, dtype=torch.int64)   # below every prime
dnum = size_q // size_p
CK = P.scheme_type.ckks
ct = rnd(2 * size_q).view(2, size_q, n)
splits = [tuple(int(x) for x in a.split("x")) for a in (sys.argv[1:] or ["16x8", "32x4", "8x16"])]
for nb, ng in splits:
    baby = [pow(5, j, 2 * n) for j in range(nb)]
    giant = [pow(5, nb * i, 2 * n) for i in range(ng)]
    def key():
        return P.PhantomRelinKey([rnd(2 * len(primes)).view(2, len(primes), n) for _ in range(dnum)])
    bk = [None] + [key() for _ in baby[1:]]
    gk = [None] + [key() for _ in giant[1:]]
    ws = [[rnd(size_q + size_p) for _ in baby] for _ in giant]
    for _ in range(2):
        out = W.diag_matvec_bsgs(ctx, size_q, ct, baby, bk, giant, gk, ws, CK)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = W.diag_matvec_bsgs(ctx, size_q, ct, baby, bk, giant, gk, ws, CK)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{nb:3d} baby x {ng:2d} giant steps ({nb * ng} diagonals, {nb + ng - 2} keys): {ms:7.3f} ms per block", flush=True)
    # 8 row blocks against the same ciphertext through the multi-block entry (8 / ng blocks share one pass over the baby keys)
    nblk = 8
    blocks = [ws] + [[[rnd(size_q + size_p) for _ in baby] for _ in giant] for _ in range(nblk - 1)]
    for _ in range(2):
        out = W.diag_matvec_bsgs_blocks(ctx, size_q, ct, baby, bk, giant, gk, blocks, CK)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = W.diag_matvec_bsgs_blocks(ctx, size_q, ct, baby, bk, giant, gk, blocks, CK)
    torch.cuda.synchronize()
    ms8 = (time.perf_counter() - t0) / 3 * 1e3 / nblk
    print(f"      the same, {nblk} row blocks per call sequence ({max(1, 16 // ng)} per pass over the keys): {ms8:7.3f} ms per block", flush=True)
    del bk, gk, ws, blocks
