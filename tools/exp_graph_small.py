"""r05: is BASELINE config 2's shape (forward + inverse NTT, N = 2^14, 8 limbs) bound by the launches or by the kernels?  The same 200 transforms
eagerly (two kernel launches each from the host) and replayed from ONE hipGraph (no host work between the kernels), event-timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

dev = torch.device("cuda:0")
for log_n, L in ((14, 8), (16, 45)):
    n = 1 << log_n
    primes = [int(p) for p in P.coeff_modulus_create(n, [50] * L)]
    ctx = P.PhantomContext(log_n, primes, 0, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(2)
    buf = torch.randint(0, 1 << 49, (16, L, n), dtype=torch.int64, device=dev, generator=g)
    REPS = 200

    def body():
        for i in range(REPS):
            ctx.nwt_2d_radix8_forward_inplace(buf[i % 16], L, 0)

    def body_inv():
        for i in range(REPS):
            ctx.nwt_2d_radix8_backward_inplace(buf[i % 16], L, 0)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / REPS * 1e3

    res = {}
    for name, fn in (("forward", body), ("inverse", body_inv)):
        eager = min(timed(fn) for _ in range(3))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()                                   # warm-up on the capture stream (arena growth is an allocation)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = min(timed(gr.replay) for _ in range(3))
        res[name] = (eager, graph)
    print(f"N = 2^{log_n}, {L} limbs: us per transform  forward eager {res['forward'][0]:.2f} / graph {res['forward'][1]:.2f}   inverse eager {res['inverse'][0]:.2f} / graph {res['inverse'][1]:.2f}", flush=True)
    del ctx, buf
