"""r05: the one-launch transforms at SMALL launch shapes.  The two-launch form costs ~10 us whatever the size; the
experiments library holds (a) both passes in one launch with the hand-off through the XCD's L2 (ntt_fused_kernel, pha_set_tuning bit 9;
lost by 7 % at 720 limbs in r02) and (b) the one-workgroup plan of N = 2^14 (WholePlan14, bit 8).  This times forward and inverse
transforms of one polynomial of L limbs (rotating over 16 buffers) for the product's choice and for those forms, and checks that every
form gives the product's words.  Run with PHA_LIB_OVERRIDE=.../libphantom_amd_exp.so.
usage: python tools/exp_onelaunch_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

DEFAULT = 1 | 32 | 64 | 2048 | 4096
dev = torch.device("cuda:0")


def t(fn, reps=300):
    for _ in range(30):
        fn(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps):
        fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for log_n, limb_list in ((14, (8, 16)), (15, (30,)), (16, (16, 32, 45))):
    n = 1 << log_n
    primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * 59)]
    ctx = P.PhantomContext(log_n, primes, 0, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    src = torch.randint(0, 1 << 49, (16, 60, n), dtype=torch.int64, device=dev, generator=gen)
    forms = [("product choice", DEFAULT, {}),
             ("two launches, plan 4 (one-wavefront contiguous pass, on-the-fly twiddles)", 1 | 16 | 64 | 1024, {}),
             ("ONE launch, L2 hand-off, lag 0", 1 | 16 | 64 | 512, {3: 0}),
             ("ONE launch, L2 hand-off, lag 1", 1 | 16 | 64 | 512, {3: 1}),
             ("ONE launch, L2 hand-off, lag 2", 1 | 16 | 64 | 512, {3: 2})]
    if log_n == 14:
        forms.append(("ONE launch, one 1024-thread workgroup per limb (WholePlan14)", DEFAULT | 256, {}))
    print(f"--- N = 2^{log_n}: us per transform (forward / inverse), one polynomial of L limbs, rotating over 16 buffers")
    for L in limb_list:
        want_f = want_i = None
        for name, variant, extra in forms:
            P.set_tuning(0, variant)
            for k, v in extra.items():
                P.set_tuning(k, v)
            buf = src.clone()
            try:
                ctx.nwt_2d_radix8_forward_inplace(buf[0], L, 0)
                f_words = buf[0, :L].clone()
                ctx.nwt_2d_radix8_backward_inplace(buf[0], L, 0)
                ok_rt = bool(torch.equal(buf[0, :L], src[0, :L]))
                if want_f is None:
                    want_f = f_words
                same = bool(torch.equal(f_words, want_f))
                f = t(lambda i: ctx.nwt_2d_radix8_forward_inplace(buf[i % 16], L, 0))
                v = t(lambda i: ctx.nwt_2d_radix8_backward_inplace(buf[i % 16], L, 0))
                print(f"L = {L:3d}  {f:7.2f} {v:7.2f}   same words {same}  round trip {ok_rt}   {name}", flush=True)
            except Exception as e:   # noqa: BLE001
                print(f"L = {L:3d}  FAILED {type(e).__name__}: {e}   {name}", flush=True)
    P.set_tuning(0, DEFAULT)
    P.set_tuning(3, 2)
    del ctx, src
