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
L.pha_exp_read_wg_times.argtypes = [C.POINTER(C.c_ulonglong)]
P.set_tuning(0, 1)
for _ in range(3):
    ctx.nwt_2d_radix8_forward_inplace(x, 45, 0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
L.pha_exp_read_wg_times(buf)
t = np.array(list(buf), dtype=np.int64).reshape(1024, 2)[:720]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) * 10 / 1000.0   # us (100 MHz ticks)
end = (t[:, 1] - t0) * 10 / 1000.0
dur = end - start
print("pass-2 kernel, 720 workgroups: start min/median/max = %.2f / %.2f / %.2f us" % (start.min(), np.median(start), start.max()))
print("end   min/median/max = %.2f / %.2f / %.2f us" % (end.min(), np.median(end), end.max()))
print("dur   min/median/max = %.2f / %.2f / %.2f us" % (dur.min(), np.median(dur), dur.max()))
hist, edges = np.histogram(start, bins=8)
print("start histogram:", list(hist), [round(e, 1) for e in edges])
# integer-path limb (limb 0 = 60-bit) vs fp limbs
print("limb0 (int path) dur median %.2f ; limbs 1..44 (fp) dur median %.2f" % (np.median(dur[:16]), np.median(dur[16:])))
