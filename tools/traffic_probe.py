"""Workload for the PMC traffic passes (tools/traffic.sh): nothing but NTT_STEPS forward NTT steps of the bench's headline
batch (16 x 45 limbs, N = 2^16) and HM_OPS single HomMul + relinearize + rescale operations, so that the per-kernel
FETCH_SIZE / WRITE_SIZE sums divide cleanly into bytes per step and bytes per op."""
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
print("probe done", NTT_STEPS, HM_OPS)
