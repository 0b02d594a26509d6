"""Timing of BASELINE config 5 on one GPU: one 128-diagonal block (pha_hoisting_weighted) at the C3 parameter set
(N = 2^16, 45 + 15 limbs) vs the same sum from single rotations + multiply_plain + add (development helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phantom_fhe_amd as P
for _kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):
    P.set_tuning(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
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
for n_diag in (16, 128):
    elts = [1] + [pow(5, k, 2 * n) for k in range(1, n_diag)]
    keys = [None] + [P.PhantomRelinKey([rnd(2 * len(primes)).view(2, len(primes), n) for _ in range(dnum)]) for _ in elts[1:]]
    diags = [rnd(size_q + size_p) for _ in elts]
    ct = rnd(2 * size_q).view(2, size_q, n)
    for _ in range(2):
        out = W.diag_matvec(ctx, size_q, ct, elts, keys, diags, CK)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = W.diag_matvec(ctx, size_q, ct, elts, keys, diags, CK)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    key_gb = (n_diag - 1) * dnum * 2 * len(primes) * n * 8 / 1e9
    print(f"{n_diag:3d} diagonals hoisted + weighted: {ms:8.3f} ms per block ({ms / n_diag:.3f} ms per diagonal; keys {key_gb:.1f} GB -> {key_gb / ms:.2f} TB/s of key traffic alone)")
    # the same sum from the reference's single-ciphertext steps: rotate (apply_galois + key switch), multiply_plain, add
    acc = torch.zeros_like(ct); g = torch.empty_like(ct); rot = torch.empty_like(ct); prod = torch.empty_like(ct)
    def naive():
        acc.zero_()
        for e, k, w in zip(elts, keys, diags):
            if e == 1:
                rot.copy_(ct)
            else:
                for p in range(2):
                    ctx.apply_galois_ntt(ct[p], g[p], e, size_q)
                rot.zero_(); rot[0].copy_(g[0])
                ctx.keyswitch_inplace(size_q, rot, g[1], k.public_keys_ptr, CK)
            for p in range(2):
                ctx.multiply_rns_poly(rot[p], w[:size_q], prod[p], size_q)
                ctx.add_rns_poly(acc[p], prod[p], acc[p], size_q)
    naive(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        naive()
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / 2 * 1e3
    print(f"{n_diag:3d} diagonals, one rotation at a time:  {ms2:8.3f} ms per block ({ms2 / n_diag:.3f} ms per diagonal) -> hoisted form is {ms2 / ms:.1f}x faster")
    del keys, diags
