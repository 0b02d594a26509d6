"""Workload for the PMC traffic passes (tools/traffic.sh) and the stage tables: nothing but NTT_STEPS forward NTT steps of the bench's
headline batch (16 x 45 limbs, N = 2^16), HM_OPS single HomMul + relinearize + rescale operations, or (r05, mode hommul_batched[:B])
5 op sets of B ciphertext pairs through the batched entry points -- so that the per-kernel FETCH_SIZE / WRITE_SIZE sums and kernel
times divide cleanly into bytes / microseconds per step and per op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

NTT_STEPS, HM_OPS = 6, 6
n, size_p = 1 << 16, 15
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44 + [60] * 15)]
size_q = 45
dev = torch.device("cuda:0")
ctx = P.PhantomContext(16, primes, size_p, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(3)
def uni(ps):
    out = torch.empty((len(ps), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(ps):
        out[i] = torch.randint(0, q, (n,), dtype=torch.int64, device=dev, generator=gen)
    return out
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("ntt", "both"):
    polys = torch.stack([uni(primes[:size_q]) for _ in range(16)])
    for _ in range(NTT_STEPS):
        ctx.nwt_2d_radix8_forward_inplace_batched(polys, size_q, 0, 16, size_q * n)
    torch.cuda.synchronize()
if mode in ("hommul", "both"):
    rlk = P.PhantomRelinKey([torch.stack([uni(primes), uni(primes)]) for _ in range(size_q // size_p)])
    ct1 = torch.stack([uni(primes[:size_q]) for _ in range(2)]); ct2 = torch.stack([uni(primes[:size_q]) for _ in range(2)])
    buf = torch.zeros((3, size_q, n), dtype=torch.int64, device=dev)
    out = torch.zeros((2, size_q - 1, n), dtype=torch.int64, device=dev)
    for _ in range(HM_OPS):     # the bench's HomMul leg: multiply out of place, key switch + rescale as one entry
        ctx.tensor_prod_2x2_rns_poly(ct1, ct2, buf, size_q)
        ctx.keyswitch_rescale(size_q, buf, buf[2], rlk.public_keys_ptr, out)
    torch.cuda.synchronize()
if mode.startswith("hommul_batched"):   # "hommul_batched[:B]": HB_SETS op sets of B ciphertext pairs through the batched entry points
    B = int(mode.split(":")[1]) if ":" in mode else 8
    HB_SETS = 5
    rnd = lambda *s: torch.randint(0, 1 << 49, s, dtype=torch.int64, device=dev, generator=gen)   # below every prime of the set
    rlk = P.PhantomRelinKey([rnd(2, len(primes), n) for _ in range(size_q // size_p)])
    bt1, bt2 = rnd(B, 2, size_q, n), rnd(B, 2, size_q, n)
    b01 = torch.zeros_like(bt1)
    b2 = torch.zeros((B, size_q, n), dtype=torch.int64, device=dev)
    bout = torch.zeros((B, 2, size_q - 1, n), dtype=torch.int64, device=dev)
    for _ in range(HB_SETS):
        ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, size_q, B)
        ctx.keyswitch_rescale_batched(size_q, b01, b2, B, rlk.public_keys_ptr, bout)
    torch.cuda.synchronize()
    print("probe done", HB_SETS, "sets of", B)
print("probe done", NTT_STEPS, HM_OPS)
