"""Experiment: one 45-limb forward NTT as C chunks of limbs on C streams (fork / join with events) against one launch pair:
do kernels of different chunks overlap their load / compute / store phases in the latency regime?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

n = 1 << 16
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 44)]
ctx = P.PhantomContext(16, primes, 0, device=0)
x = torch.randint(0, 1 << 49, (45, n), dtype=torch.int64, device="cuda:0")
main = torch.cuda.current_stream()

def timed(fn, reps=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

print("one launch pair: %.1f us" % timed(lambda: ctx.nwt_2d_radix8_forward_inplace(x, 45, 0)))
for C in (2, 3, 5):
    streams = [torch.cuda.Stream() for _ in range(C)]
    per = 45 // C
    def chunked():
        ev = torch.cuda.Event(); ev.record(main)
        done = []
        for i, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                ctx.nwt_2d_radix8_forward_inplace(x, per if i < C - 1 else 45 - per * (C - 1), i * per)
                d = torch.cuda.Event(); d.record(st); done.append(d)
        for d in done: main.wait_event(d)
    print("%d chunks on %d streams: %.1f us" % (C, C, timed(chunked)))
    def serial_chunks():
        for i in range(C):
            ctx.nwt_2d_radix8_forward_inplace(x, per if i < C - 1 else 45 - per * (C - 1), i * per)
    print("%d chunks, one stream:    %.1f us" % (C, timed(serial_chunks)))
