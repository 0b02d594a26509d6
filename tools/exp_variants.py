"""r04: time the headline step (forward NTT of 16 x 45 limbs at N = 2^16), one 45-limb polynomial and one HomMul + key switch + rescale
with several experiment builds of the library (tools/build_variant.sh), one subprocess per build (PHA_LIB_OVERRIDE is per process).
usage: python tools/exp_variants.py [name[:ENV=VAL,...] ...]   ("product" = the shipped library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "phantom-fhe_amd")

CHILD = r'''
import os, sys
sys.path[:0] = [%r, %r, %r]
import torch
import phantom_fhe_amd as P
LOG_N, SIZE_P = 16, 15
BITS = [60] + [50] * 44 + [60] * 15
n = 1 << LOG_N
dev = torch.device("cuda:0")
primes = [int(p) for p in P.coeff_modulus_create(n, BITS)]
ql = len(primes) - SIZE_P
ctx = P.PhantomContext(LOG_N, primes, SIZE_P, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
polys = torch.randint(0, 1 << 49, (16, ql, n), dtype=torch.int64, device=dev, generator=g)
ref = polys.clone()
def timed(step, steps):
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
ctx.nwt_2d_radix8_forward_inplace_batched(polys, ql, 0, 16, ql * n)
chk = int(polys.sum().item())          # checksum of one transform of the seeded input: must agree between builds
ctx.nwt_2d_radix8_backward_inplace_batched(polys, ql, 0, 16, ql * n)
rt = bool(torch.equal(polys, ref))
best = 1e9
for rep in range(3):
    best = min(best, timed(lambda: ctx.nwt_2d_radix8_forward_inplace_batched(polys, ql, 0, 16, ql * n), 60))
inv = min(timed(lambda: ctx.nwt_2d_radix8_backward_inplace_batched(polys, ql, 0, 16, ql * n), 60) for _ in range(2))
one = min(timed(lambda: ctx.nwt_2d_radix8_forward_inplace(polys[0], ql, 0), 200) for _ in range(2))
# HomMul + relinearize + rescale
evk = [torch.randint(0, 1 << 49, (2, len(primes), n), dtype=torch.int64, device=dev, generator=g) for _ in range(ql // SIZE_P)]
rlk = P.PhantomRelinKey(evk)
ct1 = torch.randint(0, 1 << 49, (2, ql, n), dtype=torch.int64, device=dev, generator=g)
ct2 = torch.randint(0, 1 << 49, (2, ql, n), dtype=torch.int64, device=dev, generator=g)
buf = torch.zeros((3, ql, n), dtype=torch.int64, device=dev)
out = torch.zeros((2, ql - 1, n), dtype=torch.int64, device=dev)
def hommul():
    ctx.tensor_prod_2x2_rns_poly(ct1, ct2, buf, ql)
    ctx.keyswitch_rescale(ql, buf, buf[2], rlk.public_keys_ptr, out)
hm = min(timed(hommul, 40) for _ in range(2))
hchk = int(out.sum().item())
print("RESULT %%8.1f us/step frac %%.3f | inverse %%8.1f | one 45-limb poly %%6.1f us | hommul %%7.1f us | roundtrip %%s chk %%x hchk %%x" %% (
      best, 16.0 * n * ql * 16 / (best * 1e-6) / 8e12, inv, one, hm, rt, chk & 0xffffffff, hchk & 0xffffffff))
''' % (ROOT, PKG, os.path.join(ROOT, "tests"))

for spec in sys.argv[1:] or ["product"]:
    name, _, envs = spec.partition(":")
    env = dict(os.environ)
    if name != "product":
        env["PHA_LIB_OVERRIDE"] = os.path.join(PKG, "phantom_fhe_amd", f"libphantom_amd_{name}.so")
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{spec:40s} {line[0][7:] if line else 'FAILED: ' + r.stderr[-300:]}", flush=True)
