"""ntt_bench equivalent (benchmark/ntt_bench.cu:104-117 sweeps batch x N): forward and inverse NTT throughput per
polynomial degree and limb count, timed with events on the launch stream (GPU box).  50-bit primes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

print("| N | limbs | forward µs | inverse µs | forward M limb-NTT/s | % of 8 TB/s (16 B x N x limbs) |")
print("|---|---|---|---|---|---|")
if os.environ.get("PHA_NTT_VARIANT"):
    P.set_tuning(0, int(os.environ["PHA_NTT_VARIANT"]))
for kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):
    P.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
LOGNS = [int(v) for v in os.environ.get("SWEEP_LOGNS", "12,13,14,15,16,17").split(",")]
for log_n in LOGNS:
    n = 1 << log_n
    primes = [int(p) for p in P.coeff_modulus_create(n, [50] * 60)]
    ctx = P.PhantomContext(log_n, primes, 0, device=0)
    for limbs, batch in ((1, 1), (8, 1), (45, 1), (60, 1), (60, 4), (60, 17), (60, 68)):   # 1, 8, 45, 60, 240, 1020, 4080 limbs per launch
        total = limbs * batch
        if total * n * 8 > (2 << 30):
            continue
        x = torch.randint(0, 1 << 49, (batch, 60, n), dtype=torch.int64, device="cuda:0")
        def run(fwd):
            f = ctx.nwt_2d_radix8_forward_inplace_batched if fwd else ctx.nwt_2d_radix8_backward_inplace_batched
            for _ in range(5):
                f(x, limbs, 0, batch, 60 * n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                f(x, limbs, 0, batch, 60 * n)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        fw, bw = run(True), run(False)
        print(f"| 2^{log_n} | {total} | {fw:.1f} | {bw:.1f} | {total / fw:.2f} | {16.0 * n * total / (fw * 1e-6) / 8e12 * 100:.1f} |")
