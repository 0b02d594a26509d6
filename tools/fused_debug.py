"""Debug aid for the one-launch NTT.  Build the instrumented library first (it is not part of the normal build):
  cd phantom-fhe_amd/csrc && for f in *.hip; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPHA_FUSED_DEBUG -c $f -o /tmp/dbg_${f%.hip}.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libphantom_dbg.so /tmp/dbg_*.o
then run with PHA_LIB_OVERRIDE=$PWD/tools/libphantom_dbg.so.  It counts workgroups
whose XCC_ID differs from blockIdx.x % 8 and polls that ran out, for eager launches and for a hipGraph replay."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P
from phantom_fhe_amd import lib as L_

L = L_.load()
def dbg(tag):
    torch.cuda.synchronize()
    out = (C.c_uint32 * 8)()
    L.pha_fused_debug_read(out)
    print(f"{tag}: xcc mismatches {out[0]} (first: block {out[1]} saw class word {out[3]}, own {out[2]}), poll timeouts {out[4]}", flush=True)

n = 1 << 16
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44)]
ctx = P.PhantomContext(16, primes, 0, device=0)
P.set_tuning(0, 1 | 32 | 64 | 512)
nb = int(os.environ.get("NB", "16"))
x = torch.randint(0, 1 << 49, (nb, 45, n), dtype=torch.int64, device="cuda:0")
step = lambda: ctx.nwt_2d_radix8_forward_inplace_batched(x, 45, 0, nb, 45 * n)
step(); dbg("first eager launch")
for _ in range(20): step()
dbg("20 eager launches back to back")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(10): step()
torch.cuda.current_stream().wait_stream(side)
dbg("after capture")
g.replay(); dbg("graph replay of 10 steps")
g.replay(); dbg("second replay")
