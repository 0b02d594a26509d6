"""r04 experiment: does the headline step (forward NTT of 16 x 45 limbs at N = 2^16, 360 MiB) get faster when the batch is
cut into chunks that fit the 256 MiB Infinity Cache (MALL), so that the contiguous pass reads what the strided pass
just wrote from the MALL instead of from HBM?

One step is always the same work (all 16 polynomials, in place, HBM-resident at its start because the whole batch is
larger than the MALL); only the launch order changes:
  whole            : one launch pair over 720 limbs (what r03 ships)
  chunks of c      : 16 / c launch pairs over 45 c limbs each, one stream
  chunks, 2 streams: the same, alternating between two streams (one chunk's tail under the next one's head)
The chunked forms use the library's own batched entry point on sub-ranges of the batch: no library change needed to measure.
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import phantom_fhe_amd as P

LOG_N, SIZE_P = 16, 15
BITS = [60] + [50] * 44 + [60] * 15
n = 1 << LOG_N
dev = torch.device("cuda:0")
primes = [int(p) for p in P.coeff_modulus_create(n, BITS)]
ql = len(primes) - SIZE_P
ctx = P.PhantomContext(LOG_N, primes, SIZE_P, device=dev)
NB = int(os.environ.get("NB", "16"))
g = torch.Generator(device=dev)
g.manual_seed(1)
polys = torch.randint(0, 1 << 49, (NB, ql, n), dtype=torch.int64, device=dev, generator=g)
stride = ql * n
STEPS = int(os.environ.get("STEPS", "60"))


def timed(step, steps=STEPS):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3   # us per step


def whole():
    ctx.nwt_2d_radix8_forward_inplace_batched(polys, ql, 0, NB, stride)


def chunked(c):
    def step():
        for lo in range(0, NB, c):
            ctx.nwt_2d_radix8_forward_inplace_batched(polys[lo:lo + c], ql, 0, min(c, NB - lo), stride)
    return step


side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def chunked_two_streams(c):
    def step():
        main = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(main)
        for i, lo in enumerate(range(0, NB, c)):
            with torch.cuda.stream(side[i & 1]):
                ctx.nwt_2d_radix8_forward_inplace_batched(polys[lo:lo + c], ql, 0, min(c, NB - lo), stride)
        for s in side:
            main.wait_stream(s)
    return step


alg = 16.0 * n * ql * NB
rows = []
for rep in range(2):
    t = timed(whole)
    rows.append(("whole (one launch pair)", t))
    for c in (8, 6, 4, 3, 2, 1):
        if c > NB:
            continue
        rows.append((f"chunks of {c} polynomials ({c * ql * n * 8 >> 20} MiB), one stream", timed(chunked(c))))
    for c in (4, 2, 1):
        rows.append((f"chunks of {c} polynomials, two streams", timed(chunked_two_streams(c))))
for name, t in rows:
    print(f"{name:62s} {t:8.1f} us/step  {alg / (t * 1e-6) / 1e12:6.3f} TB/s algorithmic  frac {alg / (t * 1e-6) / 8e12:.3f}")
