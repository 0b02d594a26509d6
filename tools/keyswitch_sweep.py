"""keyswitch_bench equivalent (benchmark/keyswitch_bench.cu:10-334: modup_bench / keyswitch_bench (= the inner product) /
moddown_bench over "Dropped Levels" 0..14 at N = 2^15, {60, 50 x 29, 60 x 15}, special_modulus_size 15, CKKS) plus the
whole keyswitch_inplace, timed with events on the launch stream (GPU box).  Bytes = SURVEY.md 8(d) algorithmic figures.
PHA_SWEEP_LOGN=16 runs the commented-out N = 2^16 set of the same file ({60, 50 x 44, 60 x 15})."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "phantom-fhe_amd"))
import torch
import phantom_fhe_amd as P

log_n = int(os.environ.get("PHA_SWEEP_LOGN", "15"))
n = 1 << log_n
size_q = 30 if log_n == 15 else 45
alpha = 15
primes = [int(p) for p in P.coeff_modulus_create(n, [60] + [50] * (size_q - 1) + [60] * alpha)]
dev = torch.device("cuda:0")
ctx = P.PhantomContext(log_n, primes, alpha, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)

def uni(ps):
    out = torch.empty((len(ps), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(ps):
        out[i] = torch.randint(0, q, (n,), dtype=torch.int64, device=dev, generator=gen)
    return out

evk = [torch.stack([uni(primes), uni(primes)]) for _ in range(size_q // alpha)]
rlk = P.PhantomRelinKey(evk)
S = P.scheme_type.ckks

def timed(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

print(f"| dropped levels | Ql | beta | mod-up µs | inner product µs | mod-down µs (one poly) | keyswitch_inplace µs | inner product TB/s (alg.) | keyswitch % of 8 TB/s (alg.) |")
print("|---|---|---|---|---|---|---|---|---|")
for dropped in range(0, 15):
    ql = size_q - dropped
    qlp = ql + alpha
    beta = ctx.beta(ql)
    c2 = uni(primes[:ql])
    ct = torch.stack([uni(primes[:ql]), uni(primes[:ql])])
    t_mod_up = torch.zeros((beta, qlp, n), dtype=torch.int64, device=dev)
    cx = torch.zeros((2, qlp, n), dtype=torch.int64, device=dev)
    out = torch.zeros((ql, n), dtype=torch.int64, device=dev)
    t_up = timed(lambda: ctx.modup(ql, t_mod_up, c2, S))
    t_ip = timed(lambda: ctx.key_switch_inner_prod(ql, cx, t_mod_up, rlk.public_keys_ptr))
    cx_keep = cx.clone()
    def down():
        cx[0].copy_(cx_keep[0])            # mod-down transforms its P limbs in place: restore them (copy included in neither column:
        ctx.moddown_from_NTT(ql, out, cx[0], S)
    t_copy = timed(lambda: cx[0].copy_(cx_keep[0]))
    t_dn = timed(down) - t_copy
    t_ks = timed(lambda: ctx.keyswitch_inplace(ql, ct, c2, rlk.public_keys_ptr, S))
    b_ip = 8.0 * n * qlp * (3 * beta + 2)
    b_ks = 8.0 * n * (ql + beta * qlp) + b_ip + 2 * 8.0 * n * (qlp + ql) + 2 * 8.0 * n * 3 * ql
    print(f"| {dropped} | {ql} | {beta} | {t_up:.1f} | {t_ip:.1f} | {t_dn:.1f} | {t_ks:.1f} | {b_ip / t_ip / 1e6:.2f} | {b_ks / t_ks / 1e6 / 8 * 100:.1f} |")
