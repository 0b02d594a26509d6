import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phantom_fhe_amd as P
from util import primes_of, rng_for, uniform_poly
log_n, primes, size_p = primes_of("c3_ckks16")
n = 1 << log_n
ctx = P.PhantomContext(log_n, list(primes), size_p, device=0)
x = P.to_device(uniform_poly(rng_for(1), primes[:45], n), "cuda:0")
L = P.load()
L.pha_exp_read_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
for variant in (1, 9):
    P.set_tuning(0, variant)
    for limbs in (2, 45):
        for which in ("fwd",):
            for _ in range(3):
                ctx.nwt_2d_radix8_forward_inplace(x, limbs, 0)
            torch.cuda.synchronize()
            st = (C.c_ulonglong * 8)()
            L.pha_exp_read_stamps(st)
            s = list(st)
            # last kernel launched = pass 2 (contiguous); stamps are from it
            d = [s[i + 1] - s[i] for i in range(6)]
            print(f"v{variant} limbs={limbs} pass2 stamps (cycles): load+roundA={d[0]} barrier={d[1]} roundB={d[2]} barrier={d[3]} roundC={d[4]} drain={d[5]} total={s[6]-s[0]}")
