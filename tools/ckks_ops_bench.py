"""ckks_bench equivalent (benchmark/ckks_bench.cu:168-205 times multiply / relinearize / rescale_to_next / rotate
separately, CUDATimer: include/cuda_wrapper.cuh:191-283): the same operations of the hot path at the C3 set
(N = 2^16, 45 + 15 limbs; PHA_OPS_LOGN=15 for the C4 set), >= 100 iterations after 10 warm-ups, median + mean from event
pairs on the launch stream."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

for kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):   # experiments library only (PHA_LIB_OVERRIDE)
    P.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
log_n = int(os.environ.get("PHA_OPS_LOGN", "16"))
n = 1 << log_n
size_q = 45 if log_n == 16 else 30
alpha = 15
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * (size_q - 1) + [60] * alpha)]
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, primes, alpha, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(11)

def uni(ps):
    out = torch.empty((len(ps), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(ps):
        out[i] = torch.randint(0, q, (n,), dtype=torch.int64, device=dev, generator=gen)
    return out

def key():
    return P.PhantomRelinKey([torch.stack([uni(primes), uni(primes)]) for _ in range(size_q // alpha)])

rlk, glk = key(), key()
S = P.scheme_type.ckks
q = primes[:size_q]
ct1 = torch.stack([uni(q), uni(q)]); ct2 = torch.stack([uni(q), uni(q)])
ct3 = torch.stack([uni(q), uni(q), uni(q)])
out3 = torch.zeros_like(ct3)
resc = torch.zeros((2, size_q - 1, n), dtype=torch.int64, device=dev)
g = torch.zeros_like(ct1); rot = torch.zeros_like(ct1)
elt = 5

def stats(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    return statistics.median(us), statistics.fmean(us)

def multiply():           # multiply_inplace (ckks_multiply src/evaluate.cu:346-402): tensor product into a size-3 ciphertext
    ctx.tensor_prod_2x2_rns_poly(ct1, ct2, out3, size_q)
def relinearize():        # relinearize_inplace :1028-1077
    ctx.keyswitch_inplace(size_q, ct3, ct3[2], rlk.public_keys_ptr, S)
def rescale():            # rescale_to_next_inplace :1376-1427
    ctx.divide_and_round_q_last_ntt(size_q, ct1, 2, resc)
def relin_rescale():      # relinearize_inplace + rescale_to_next_inplace as one entry point (build-defined fusion, same result)
    ctx.keyswitch_rescale(size_q, ct3, ct3[2], rlk.public_keys_ptr, resc)
def hommul():             # multiply + relinearize + rescale
    ctx.tensor_prod_2x2_rns_poly(ct1, ct2, out3, size_q); ctx.keyswitch_rescale(size_q, out3, out3[2], rlk.public_keys_ptr, resc)
def rotate():             # rotate_inplace -> apply_galois_inplace :1567-1624 (NTT-domain permutation + key switch)
    ctx.apply_galois_ntt(ct1[0], g[0], elt, size_q); ctx.apply_galois_ntt(ct1[1], g[1], elt, size_q)
    rot[0].copy_(g[0]); rot[1].zero_()
    ctx.keyswitch_inplace(size_q, rot, g[1], glk.public_keys_ptr, S)
def add():                # add_inplace :116-198
    ctx.add_rns_poly(ct1, ct2, g, 2 * size_q) if False else (ctx.add_rns_poly(ct1[0], ct2[0], g[0], size_q), ctx.add_rns_poly(ct1[1], ct2[1], g[1], size_q))

print(f"CKKS N = 2^{log_n}, {size_q} + {alpha} limbs, one MI355X; 100 iterations after 10 warm-ups (event pairs on the launch stream)")
print("| operation | median µs | mean µs |")
print("|---|---|---|")
for name, fn in (("multiply (tensor product)", multiply), ("relinearize (key switch)", relinearize), ("rescale_to_next", rescale),
                 ("relinearize + rescale, one entry (pha_keyswitch_rescale)", relin_rescale), ("multiply + relinearize + rescale", hommul),
                 ("rotate (Galois + key switch)", rotate), ("add", add)):
    med, mean = stats(fn)
    print(f"| {name} | {med:.1f} | {mean:.1f} |")
